python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import load_keys
from vtoonify_amd import synth, _lib
from vtoonify_amd.engine import VToonifyEngine
_lib.use_library(_lib.DEFAULT_LIB)
dev = torch.device('cuda:0')
sd = {k: v.to(dev) for k, v in synth.synth_state_dict(load_keys('D'), 0).items()}
s = synth.synth_style(seed=17).to(dev)
for B in (1, 4):
    x = synth.synth_frames(B, 256, 256, seed=5).to(dev)
    for name, kw in (('fp32', {}), ('f32x3', {'x3': True})):
        eng = VToonifyEngine(sd, 'dualstylegan', 256, torch.float32, dev, **kw)
        for _ in range(3): y = eng.forward(x, s, 0.5, shared_style=True, borrow=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): y = eng.forward(x, s, 0.5, shared_style=True, borrow=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"B={B} {name}: {1e3*dt:.3f} ms/step  {B/dt:.1f} frames/s (one step in flight)", flush=True)
        if name == 'f32x3':
            plan = eng.plan_for(B, 256, 256, True, True)
            rows = {}
            for info, ms in eng.time_ops(plan, 3):
                k = info.get('kernel', '?'); rows[k] = rows.get(k, 0) + ms
            for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:10]: print(f"     {k:<40} {v:.3f} ms")
        del eng
PY
CB="python tools/conv_bench.py"
echo "--- f32x3 tile choice on the 128-channel layers (fp32 dtype... conv_bench has no x3 flag: skipped)"
