"""Build-time check of the attention kernel's hand-counted index loads (run on the build box, no GPU):

    python tools/check_attn_asm.py [-DMMEE_DIAG]

The index words of a key tile are fetched by inline-asm loads whose destination registers are carried around the tile loop and only
become valid at a hand-placed s_waitcnt (attention_idx.hip).  hipcc does not know the loads are pending, so a register copy, spill or
re-definition of those registers that it inserts anywhere would silently read or clobber words in flight.  This script compiles the file
to assembly and, for every attention_idx_kernel instantiation, verifies on the kernel's control-flow graph that
  * every index load of the kernel writes the SAME sixteen registers (no copy is needed on the loop's back edge), and
  * on every path from an index-load group to the first s_waitcnt vmcnt(N <= 4) -- the wait that covers the loads; only the four DMA
    pieces issued behind them may still be in flight -- no instruction reads or writes one of those registers.
Exit status 1 on a violation.  tests/test_host.py runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "multi-modal-early-exit_amd", "csrc", "attention_idx.hip")


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main():
    extra = [x for x in sys.argv[1:] if x.startswith("-D")]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "a.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                        "-I" + os.path.dirname(SRC)] + extra + [SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    bad = 0
    kernels = 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN4mmee20attention_idx_kernel\w+):", text[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        if "kernelILi2E" in name:                 # MODE 2 = timing variants of the diagnostic library: wrong results by design, not checked
            i += 1
            continue
        j = i + 1
        while j < len(text) and not text[j].startswith(".Lfunc_end"):
            j += 1
        # instruction list with label positions
        ins, labels = [], {}
        for l in text[i + 1:j]:
            l = l.split(";")[0].strip()
            if not l or l.startswith(";"):
                continue
            if l.endswith(":"):
                labels[l[:-1]] = len(ins)
                continue
            if l.startswith("."):
                continue
            ins.append(l)
        i = j
        is_idx = [l.startswith("global_load_dwordx4") and re.search(r", s\[\d+:\d+\]", l) is not None for l in ins]
        if not any(is_idx):
            continue
        kernels += 1
        idx = set()
        for k, l in enumerate(ins):
            if is_idx[k]:
                idx |= regs_of(l.split()[1].rstrip(","))
        if len(idx) != 16:
            print(f"{name}: the index loads write {len(idx)} registers, not one set of 16: a copy would be needed on some path")
            bad += 1
            continue

        def succ(k):
            op = ins[k].split()[0]
            if op in ("s_endpgm",):
                return []
            if op == "s_branch":
                return [labels[ins[k].split()[1]]]
            out = [k + 1] if k + 1 < len(ins) else []
            if op.startswith("s_cbranch"):
                out.append(labels[ins[k].split()[1]])
            return out

        # from the end of every index-load group, walk every path until a wait that covers the loads (vmcnt <= 4: at most the four
        # DMA pieces issued after them stay in flight); no instruction on the way may touch the sixteen registers
        ends = [k for k in range(len(ins)) if is_idx[k] and not (k + 1 < len(ins) and is_idx[k + 1])]
        viol = set()
        for e in ends:
            seen, todo = set(), list(succ(e))
            while todo:
                k = todo.pop()
                if k in seen:
                    continue
                seen.add(k)
                l = ins[k]
                mw = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
                if mw and int(mw.group(1)) <= 4:
                    continue
                used = set()
                for t in l.replace(",", " ").split()[1:]:
                    used |= regs_of(t)
                if used & idx:
                    viol.add(l)
                todo += succ(k)
        for l in sorted(viol):
            print(f"{name}: `{l}` touches an index register while the loads are in flight")
        bad += len(viol)
        if not viol:
            print(f"{name}: {len(ends)} index-load groups -> v{min(idx)}..v{max(idx)}, untouched until the covering wait on every path: ok")
    if not kernels:
        print("no attention_idx_kernel with index loads found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
