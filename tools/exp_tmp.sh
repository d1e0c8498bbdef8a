python - <<'PY'
import sys, torch, numpy as np, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_keys, load_golden
from vtoonify_amd import synth, _lib, kernels as K
from vtoonify_amd.engine import VToonifyEngine
_lib.use_library(_lib.DEFAULT_LIB)
dev = torch.device('cuda:0')
sd = {k: v.to(dev) for k, v in synth.synth_state_dict(load_keys('D'), 0).items()}
d, _ = load_golden("e2e_D.npz")
x, s = torch.from_numpy(d["x"]).to(dev), torch.from_numpy(d["style"]).to(dev)
done = False
for trial in range(30):
    if done and trial > 20: break
    eng = VToonifyEngine(sd, 'dualstylegan', 256, torch.float32, dev, x3=True)
    def run(ops, stream, plan=None, eng=eng):
        global done
        pl = list(eng._plans.values())[-1]
        for i, (fn, args, what) in enumerate(ops):
            info = what if isinstance(what, dict) else {"name": what}
            before = {n for n, t in pl.bufs.items() if t.is_floating_point() and bool(torch.isnan(t.float()).any())}
            fn(*args, stream)
            torch.cuda.synchronize()
            after = {n for n, t in pl.bufs.items() if t.is_floating_point() and bool(torch.isnan(t.float()).any())}
            new = after - before
            if new and info.get("name") == "conv" and not before:
                done = True
                name = sorted(new)[0]
                t = pl.bufs[name]
                print(f"trial {trial} op {i} {info.get('kernel')} sig={info.get('sig')}: NEW NaN in {sorted(new)} ({int(torch.isnan(t.float()).sum())} of {t.numel()}); before: {sorted(before)}")
                nn = torch.isnan(t.float()).nonzero()
                print("   NaN index min/max per dim:", nn.min(0).values.tolist(), nn.max(0).values.tolist(), "distinct channels", sorted(set(nn[:, -1].tolist()))[:70])
                desc = args[0]._obj
                print("   desc: n,h,w", desc.n, desc.h, desc.w, "c0,c1", desc.c0, desc.c1, "cout", desc.cout, "dil", desc.dil, "dtype", desc.dtype, "resid", bool(desc.resid), "hint", desc.tile_hint, "tile", eng.lib.vt_conv2d_tile(C.byref(desc)))
                for rep in range(4):
                    fn(*args, stream); torch.cuda.synchronize()
                    print(f"   rerun {rep}: NaNs {int(torch.isnan(t.float()).sum())}", end="")
                    nz = t.float(); print("  finite max", float(nz[~torch.isnan(nz)].abs().max()) if (~torch.isnan(nz)).any() else None)
                desc.dtype = K.VT_F32
                fn(*args, stream); torch.cuda.synchronize()
                print(f"   exact fp32 rerun: NaNs {int(torch.isnan(t.float()).sum())}")
                desc.dtype = K.VT_F32X3
                fn(*args, stream); torch.cuda.synchronize()
                print(f"   f32x3 again: NaNs {int(torch.isnan(t.float()).sum())}")
                idx = torch.isnan(t.float()).nonzero()[:5].tolist() if torch.isnan(t.float()).any() else []
                print("   first NaN indices", idx, "shape", tuple(t.shape))
    eng._run = run
    y = eng.forward(x, s, 0.5, use_graph=False)
    print(f"trial {trial}: nan in y {bool(torch.isnan(y).any())}", flush=True)
    del eng
PY
