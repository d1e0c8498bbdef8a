import re, sys
lines = open(sys.argv[1]).read().split('\n')
pend = []  # list of (set(regs), line_no) in issue order (incl. lds dma as empty set)
def regs_of(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    if m: return {int(m.group(1))}
    return set()
bad = 0
for i, l in enumerate(lines):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    toks = re.findall(r'v\[\d+:\d+\]|v\d+', t)
    if op.startswith('global_load') or (op.startswith('buffer_load') and ' lds' not in t):
        dst = regs_of(toks[0]) if toks else set()
        srcs = set().union(*[regs_of(x) for x in toks[1:]]) if len(toks) > 1 else set()
        for r, ln in pend:
            if r & srcs: print(f"line {i+1}: {t}  READS pending load (line {ln})"); bad += 1
        pend.append((dst, i + 1)); continue
    if op.startswith('buffer_load') or op.startswith('buffer_store') or op.startswith('global_store'):
        pend.append((set(), i + 1)); continue
    m = re.search(r'vmcnt\((\d+)\)', t)
    if op == 's_waitcnt' and m:
        n = int(m.group(1)); pend = pend[len(pend) - n:] if n < len(pend) else pend
        if n == 0: pend = []
        continue
    if op in ('s_cbranch_scc1','s_cbranch_scc0','s_cbranch_vccnz','s_cbranch_vccz','s_cbranch_execz','s_cbranch_execnz','s_branch'):
        continue
    allregs = set().union(*[regs_of(x) for x in toks]) if toks else set()
    for r, ln in pend:
        if r & allregs:
            print(f"line {i+1}: {t}   TOUCHES regs of pending load issued at line {ln}"); bad += 1
print("violations:", bad)
