"""Build-time check of the attention kernel's hand-counted asm loads (run on the build box, no GPU):

    python tools/check_attn_asm.py [-DMMEE_DIAG]

attention_idx.hip issues two kinds of vector loads as inline asm so that hipcc places no `s_waitcnt vmcnt(0)` of its own beside the
pending LDS-DMA pieces: the PAIR-INDEX words of a key tile (four `global_load_dwordx4` with a scalar base, 16 registers carried around
the tile loop; two loads / 8 registers in the IDX = 16 instantiations of round 6) and the Q fragments of an item (eight `global_load_dwordx4 ... off`, 32 registers).  Their destination registers only
become valid at a hand-placed counted wait.  hipcc does not know the loads are pending, so a register copy, spill, move to an AGPR or
re-definition that it inserts between the load and that wait would silently read or clobber data in flight (seen once for the index
words: a v_mov in front of the wait).  This script compiles the file to assembly WITH THE MAKEFILE'S FLAGS (`make print-flags`) and, for
every attention_idx_kernel instantiation the release / diagnostic library runs (the timing variants, MODE 2, are wrong by design and are
skipped), verifies on the kernel's control-flow graph that
  * every index load of the kernel writes the SAME sixteen registers (no copy is needed on the loop's back edge), and
  * on every path from a load group to a wait that COVERS it, no instruction reads or writes one of the group's registers.  A wait
    `s_waitcnt vmcnt(N)` covers the group when N <= 4 (the tile protocol of the source: behind a group at most the four LDS-DMA pieces
    of the next key tile are issued before the wait that names them; the control-flow graph also contains paths the source cannot
    take, e.g. "no piece issued, then the wait of a tile that has a successor", so the count alone would cry wolf) or when at least N
    vector-memory operations (loads, stores, atomics, LDS-DMA: they retire in issue order) were issued behind the group on that
    path; an `s_endpgm` ends a path.
Exit status 1 on a violation.  tests/test_host.py runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multi-modal-early-exit_amd", "csrc")
SRC = os.path.join(CSRC, "attention_idx.hip")
CAP = 40                      # vector-memory operations counted behind a group (more than any wait immediate in the kernel)


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def is_vmem(l):
    op = l.split()[0]
    return op.startswith(("global_", "buffer_", "flat_", "scratch_"))


def makefile_flags(extra):
    r = subprocess.run(["make", "-s", "-C", CSRC, "print-flags", "EXTRA=" + " ".join(extra)], capture_output=True, text=True, check=True)
    return [f for f in r.stdout.split() if f not in ("-fPIC",)]


def check_groups(name, ins, labels, groups, what):
    """groups: list of (last instruction index of the group, register set).  Returns the violating instructions."""

    def succ(k):
        op = ins[k].split()[0]
        if op == "s_endpgm":
            return []
        if op == "s_branch":
            return [labels[ins[k].split()[1]]]
        out = [k + 1] if k + 1 < len(ins) else []
        if op.startswith("s_cbranch"):
            out.append(labels[ins[k].split()[1]])
        return out

    viol = set()
    for end, regs in groups:
        seen, todo = set(), [(k, 0) for k in succ(end)]
        while todo:
            k, cnt = todo.pop()
            if (k, cnt) in seen:
                continue
            seen.add((k, cnt))
            l = ins[k]
            mw = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if mw and int(mw.group(1)) <= max(cnt, 4):
                continue                                       # the group has landed on this path
            used = set()
            for t in l.replace(",", " ").split()[1:]:
                used |= regs_of(t)
            if used & regs:
                viol.add(l)
                continue
            c2 = min(CAP, cnt + 1) if is_vmem(l) else cnt
            todo += [(s, c2) for s in succ(k)]
    for l in sorted(viol):
        print(f"{name}: `{l}` touches a register of {what} while the loads are in flight")
    return len(viol)


def main():
    extra = [x for x in sys.argv[1:] if x.startswith("-D")]
    flags = makefile_flags(extra)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "a.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-I" + CSRC, SRC, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
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
        ins, labels = [], {}
        for l in text[i + 1:j]:
            l = l.split(";")[0].strip()
            if not l:
                continue
            if l.endswith(":"):
                labels[l[:-1]] = len(ins)
                continue
            if l.startswith("."):
                continue
            ins.append(l)
        i = j
        kernels += 1
        is_ld = [l.startswith("global_load_dwordx4") for l in ins]
        is_idx = [is_ld[k] and re.search(r", s\[\d+:\d+\]", ins[k]) is not None for k in range(len(ins))]
        is_q = [is_ld[k] and re.search(r", off\b", ins[k]) is not None for k in range(len(ins))]
        dest = lambda k: regs_of(ins[k].split()[1].rstrip(","))
        # ---- pair-index loads (BIAS kernels): one set of 16 registers, groups of four
        n_idx = 0
        if any(is_idx):
            idx = set()
            for k in range(len(ins)):
                if is_idx[k]:
                    idx |= dest(k)
            want = 8 if re.search(r"ELi16(ELb[01])?EEEvNS_8AttnArgsEPyi$", name) else 16      # IDX = 16 (round 6): two loads per tile, eight registers
            if len(idx) != want:
                print(f"{name}: the index loads write {len(idx)} registers, not one set of {want}: a copy would be needed on some path")
                bad += 1
                continue
            ends = [k for k in range(len(ins)) if is_idx[k] and not (k + 1 < len(ins) and is_idx[k + 1])]
            n_idx = len(ends)
            bad_here = check_groups(name, ins, labels, [(e, idx) for e in ends], "the pair-index words")
            bad += bad_here
            if not bad_here:
                print(f"{name}: {n_idx} index-load groups -> v{min(idx)}..v{max(idx)}, untouched until a covering wait on every path: ok")
        # ---- Q fragment loads (every kernel): runs of eight `global_load_dwordx4 v[..], v[..], off`
        groups = []
        k = 0
        while k < len(ins):
            if is_q[k]:
                e = k
                regs = set()
                while e < len(ins) and is_q[e]:
                    regs |= dest(e)
                    e += 1
                if e - k == 8:
                    groups.append((e - 1, regs))
                k = e
            else:
                k += 1
        if not groups:
            print(f"{name}: no group of eight Q loads found (the kernel's asm changed: update this check)")
            bad += 1
            continue
        if any(len(r) != 32 for _, r in groups):
            print(f"{name}: a Q-load group does not write 32 distinct registers")
            bad += 1
            continue
        bad_here = check_groups(name, ins, labels, groups, "the Q fragments")
        bad += bad_here
        if not bad_here:
            print(f"{name}: {len(groups)} Q-load group(s), 32 registers each, untouched until a covering wait on every path: ok")
    if not kernels:
        print("no attention_idx_kernel found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
