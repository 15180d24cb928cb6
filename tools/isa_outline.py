"""Outline of a kernel's instruction stream from hipcc's -S output: run-length classes (M mfma, R ds_read, W ds_write, D buffer/global load,
S store, B s_barrier, w[…] s_waitcnt, P s_setprio, v / s other vector / scalar) — to check that a hand-placed schedule survived the compiler.
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/k.s file.hip;  python tools/isa_outline.py /tmp/k.s <mangled-name-substring> [label]"""
import re
import sys

s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r"^(\S+):\s*; @", s, re.M) if sys.argv[2] in m.group(1)]
for name in names:
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    out = []
    for ln in s[i:j].splitlines():
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.startswith(".LBB"):
                out.append("\n" + t.split(":")[0] + ":")
            continue
        op = t.split()[0]
        k = ("M" if op.startswith("v_mfma") else "R" if op.startswith("ds_read") else "W" if op.startswith("ds_write") else
             "D" if op.startswith(("buffer_load", "global_load")) else "S" if op.startswith(("buffer_store", "global_store")) else
             "B" if op == "s_barrier" else "w[" + t.split(None, 1)[1] + "]" if op == "s_waitcnt" else "P" if op == "s_setprio" else
             "br" if op.startswith(("s_cbranch", "s_branch")) else "v" if op.startswith("v_") else "s")
        out.append(k)
    res, prev, cnt = [], None, 0
    for k in out + [None]:
        if k == prev:
            cnt += 1
        else:
            if prev is not None:
                res.append(prev + (str(cnt) if cnt > 1 else ""))
            prev, cnt = k, 1
    text = " ".join(res)
    if len(sys.argv) > 3:
        a = text.find(sys.argv[3])
        text = text[a:a + 6000]
    print(name, "\n", text)
