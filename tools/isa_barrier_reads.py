"""For every kernel in a gfx950 assembly listing: LDS reads that may still be in flight (issued, not yet covered by an `s_waitcnt lgkmcnt`) when
the wave reaches an `s_barrier` - a dataflow over the kernel's basic blocks (maximum over the predecessors, to a fixed point).
A barrier that releases an LDS buffer for refill must not be passed with fragment reads of that buffer in flight: the next tile's LDS-DMA
pieces that are range-checked away (zero fill, no memory round trip) can land before a read that waits in a contended LDS queue.  That is the
row-reuse kernel's co-scheduling hazard (profiles/r04_r3_bisect.txt): the compiler sank fragment reads and their MFMAs below the next
step's barrier in its three-stage instantiations.  Nothing in the language forbids that (`__builtin_amdgcn_s_barrier` is not a memory fence and an
asm "memory" clobber orders issue, not completion), so the product's k-step states `s_waitcnt lgkmcnt(0)` before its barrier and this scan checks
what the compiler made of every kernel.
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s kernels/conv_igemm.hip ; python tools/isa_barrier_reads.py k.s"""
import re
import sys

CAP = 64


def blocks_of(body):
    """[(label, [instructions])], successors by label / fall-through"""
    blocks, cur, name = [], [], "entry"
    for line in body.split("\n"):
        l = line.strip()
        if not l or l.startswith(";"):
            continue
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        if l.startswith("."):
            continue
        l = l.split(";")[0].strip()
        cur.append(l)
        if re.match(r"s_cbranch_", l):   # a conditional branch ends its block: what follows it does not happen on the taken edge
            blocks.append((name, cur))
            name, cur = f"{name}+{len(blocks)}", []
    blocks.append((name, cur))
    return blocks


def scan_kernel(body):
    blocks = blocks_of(body)
    index = {name: i for i, (name, _) in enumerate(blocks)}
    succ = []
    for i, (_, ins) in enumerate(blocks):
        s = []
        fall = True
        for l in ins:
            m = re.match(r"s_cbranch_\S+\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in index:
                s.append(index[m.group(1)])
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                if m.group(1) in index:
                    s.append(index[m.group(1)])
                fall = False
            if l.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    IN = [0] * len(blocks)
    flagged = {}
    changed = True
    while changed:
        changed = False
        for i, (_, ins) in enumerate(blocks):
            p = IN[i]
            for k, l in enumerate(ins):
                if re.match(r"ds_(read|load)", l):
                    p = min(p + 1, CAP)
                elif l.startswith("s_waitcnt"):
                    w = re.search(r"lgkmcnt\((\d+)\)", l)
                    if w:
                        p = min(p, int(w.group(1)))
                elif l.startswith("s_barrier") and p:
                    flagged[(i, k)] = max(flagged.get((i, k), 0), p)
                elif re.match(r"s_branch\s", l) or l.startswith("s_endpgm"):
                    break
            for j in succ[i]:
                if p > IN[j]:
                    IN[j] = p
                    changed = True
    return sorted(flagged.values())


def scan(path):
    txt = open(path).read()
    out = []
    names = re.findall(r"\.amdhsa_kernel (\S+)", txt)
    for name in names:
        m = re.search(r"\n" + re.escape(name) + r":.*?\n(.*?)\n\.Lfunc_end\d+:", txt, flags=re.S)
        if m:
            f = scan_kernel(m.group(1))
            if f:
                out.append((name, f))
    return len(names), out


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        n, bad = scan(p)
        print(f"{p}: {n} kernels, {len(bad)} with a barrier reached while LDS reads may be in flight")
        for name, ks in bad:
            print("   ", name[:150], "reads in flight at the flagged barriers:", ks)
            rc = 1
    sys.exit(rc)
