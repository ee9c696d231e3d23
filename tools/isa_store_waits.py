"""Loops whose body holds a global / buffer STORE and a full `s_waitcnt vmcnt(0)`: every trip then waits for the previous trip's store to be
acknowledged (the wait is usually the compiler's, for some load in the body) - the pattern that cost the implicit-GEMM epilogue 15 % of its time
(launch anatomy).  A listing-order approximation: a loop is the span between a label and a later branch back to it; inner spans
are reported, not their enclosing ones.  Candidates to look at, not verdicts.
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s file.hip ; python tools/isa_store_waits.py k.s"""
import re
import sys


def scan(path):
    txt = open(path).read()
    out = []
    for name in re.findall(r"\.amdhsa_kernel (\S+)", txt):
        m = re.search(r"\n" + re.escape(name) + r":.*?\n(.*?)\n\.Lfunc_end\d+:", txt, flags=re.S)
        if not m:
            continue
        lines = [l.strip().split(";")[0].strip() for l in m.group(1).split("\n")]
        lines = [l for l in lines if l]
        pos = {}
        for i, l in enumerate(lines):
            mm = re.match(r"(\.LBB\d+_\d+):", l)
            if mm:
                pos[mm.group(1)] = i
        spans = []
        for i, l in enumerate(lines):
            mm = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in pos and pos[mm.group(1)] < i:
                spans.append((pos[mm.group(1)], i))
        inner = [s for s in spans if not any(t != s and s[0] <= t[0] and t[1] <= s[1] for t in spans)]
        for a, b in inner:
            body = lines[a:b + 1]
            stores = sum(1 for l in body if re.match(r"(global_store|buffer_store|flat_store)", l))
            loads = sum(1 for l in body if re.match(r"(global_load|buffer_load|flat_load)", l) and " lds" not in l)
            waits = sum(1 for l in body if re.match(r"s_waitcnt.*vmcnt\(0\)", l))
            if stores and waits:
                out.append((name, b - a + 1, stores, loads, waits))
    return out


if __name__ == "__main__":
    for p in sys.argv[1:]:
        res = scan(p)
        print(f"{p}: {len(res)} loops with a store and a vmcnt(0) wait in the body")
        for name, n, st, ld, w in res:
            short = re.sub(r"^_ZN4trtx12_GLOBAL__N_1\d+", "", name)[:110]
            print(f"    {short:110s} {n:5d} instr  stores {st}  loads {ld}  vmcnt(0) {w}")
