"""ISA lint for the hand-counted waits of the library's kernels (no GPU needed; hipcc cross-compiles).

Two families of hand-placed `s_waitcnt` exist, and in both the compiler does not know what the wait is for:

(1) LDS reads issued as inline asm (attention_pipe.hip / attention.hip: `ds_read_b64_tr_b16`, attention_device.h lds_read_v_step) and waited for with
    hand-counted `s_waitcnt lgkmcnt(N)` asm statements.  hipcc believes the asm's output registers are written when the statement issues; it is therefore
    free to COPY them (v_mov, at a control-flow merge or when it splits a live range) before the wait -- the copy then holds whatever the register held
    before.  Round 4 hit exactly that: a branch between the reads and their wait made hipcc move 16 fragment registers 85 instructions after the reads;
    beside another kernel's LDS traffic one wave in ~10^5 read stale fragments (ViT-L/4 batch 8: 5 % of the forwards wrong in one sample).
    Rules, per kernel (`lint_isa`):
      a. between an asm `ds_read_b64_tr_b16 vX` and the first `s_waitcnt lgkmcnt(N)` that covers it (LDS returns in order: a wait for N leaves the N
         youngest LDS operations outstanding -- compiler-issued ones count too), no instruction may read vX or write it;
      b. no branch and no label (control-flow split or merge) while such a read is outstanding: that is where a compiler decides to move registers.

(2) LDS-DMA (`global_load_lds_dwordx4`) kept in flight across raw `s_barrier`s and retired with a counted `s_waitcnt vmcnt(N)` (gemm.hip: the 8-phase
    kernel's `vmcnt(6)`, the deep-ring kernel's `vmcnt(2 PER)` / `vmcnt(PER)`).  N counts INSTRUCTIONS: the wait is right only if the loop issues
    exactly the vector-memory instructions the source counted -- a spill (`scratch_store`), a hoisted or duplicated piece, or any other load / store
    between the pieces and the wait shifts which piece the wait retires, silently.
    Rules, per kernel with a counted asm wait inside a loop (`lint_vmcnt`):
      a. every vector-memory instruction of that loop is a `global_load_lds_dwordx4`;
      b. on every path once around the loop that passes a counted wait, the number of pieces issued, the number issued BEFORE the wait and the wait's
         immediate are the ones the kernel's template arguments imply (VMCNT_SPECS);
      c. the straight-line prologue waits (`vmcnt(6)` after 14 pieces) likewise.

    python tools/asm_lds_lint.py            # compiles attention_pipe.hip, attention.hip and gemm.hip; exits 1 on a hit
    python tools/asm_lds_lint.py --record   # the same, and on success writes csrc/LINT_PASSED.json {hipcc version}: build.py warns when
                                            # the compiler of a build is not the one recorded there, bench.py carries the comparison in its line
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("attention_pipe.hip", "attention.hip", "gemm.hip")


def _regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def _is_label(l):
    return bool(re.match(r"^\.LBB\d+_\d+:", l))


def _is_branch(op):
    return op.startswith("s_cbranch") or op == "s_branch" or op == "s_setpc_b64"


def lint_isa(text):
    """[(kernel, line number, instruction, register, lines since the read)] for every premature use of an asm-requested LDS register and every branch /
    label met while one is outstanding (register -1)."""
    hits, kernel, order, in_asm = [], None, [], False  # order: outstanding LDS operations, oldest first: (line, regs of an asm read or None)

    def pending():
        p = {}
        for ln, regs in order:
            for r in regs or ():
                p[r] = ln
        return p

    for i, raw in enumerate(text.split("\n")):
        l = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, order, in_asm = m.group(1), [], False
            continue
        if kernel is None or not l:
            continue
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if _is_label(l):
            p = pending()
            if p:
                hits.append((kernel, i + 1, l.split()[0] + "   [label with asm LDS reads outstanding]", -1, i - min(p.values())))
            continue
        if l[0] in ";.":
            continue
        op = l.split()[0]
        args = [a for a in l[len(op):].split(",")]
        if _is_branch(op):
            p = pending()
            if p:
                hits.append((kernel, i + 1, l + "   [branch with asm LDS reads outstanding]", -1, i - min(p.values())))
            if op == "s_setpc_b64":
                order = []
            continue
        if op == "s_endpgm":
            order = []
            continue
        if op.startswith("ds_"):
            order.append((i, _regs(args[0]) if (in_asm and op == "ds_read_b64_tr_b16") else None))
            if order[-1][1] is not None:
                continue
        if op == "s_waitcnt" and "lgkmcnt" in l:
            n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
            order = order[len(order) - n:] if n > 0 else []
            continue
        p = pending()
        if not p:
            continue
        stores = op.startswith("global_store") or op.startswith("ds_write") or op.startswith("buffer_store") or op.startswith("scratch_store")
        for a in (args if stores else args[1:]):
            for r in _regs(a):
                if r in p:
                    hits.append((kernel, i + 1, l, r, i - p[r]))
        if not stores and not op.startswith("s_"):  # ... nor may anything else be written there (the late LDS data would land on top of it)
            for r in _regs(args[0]):
                if r in p:
                    hits.append((kernel, i + 1, l + "   [overwrites]", r, i - p[r]))
    return hits


# ---- (2) counted vmcnt waits over LDS-DMA -------------------------------------------------------------------------------------------------------
def _vm_op(op):
    """Does this instruction increment vmcnt on gfx9?  (loads, stores and atomics of the vector-memory path, scratch included)"""
    return op.startswith(("global_", "buffer_", "scratch_", "flat_", "tbuffer_")) and not op.startswith("buffer_wbl2") and not op.startswith("buffer_inv")


def vmcnt_spec(kernel):
    """{immediate: (pieces per loop iteration, pieces issued before the wait in its iteration)} for the counted loop waits the kernel's source places,
    and [(pieces, immediate)] for its straight-line prologue wait; None for kernels without counted waits."""
    m = re.search(r"gemm8p_kernelILi(\d)E", kernel)
    if m:  # P1: A1(t+1)  P2: A0(t+2)  P3: W0(t+2)  P4: W1(t+2), two pieces per wave each; vmcnt(6) leaves the three youngest half-tiles in flight.
        # Half-width column tiles (round 5) skip W1: three half-tiles per K tile, vmcnt(4) leaves the two youngest in flight
        return {"loop": {6: (8, 8), 4: (6, 6)}, "prologue": [(14, 6), (10, 4)]}
    m = re.search(r"gemm_bf16_kernelILi(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", kernel)
    if m:
        _, bm, bn, wm, wn, stages = (int(x) for x in m.groups())
        if stages <= 2:
            return None
        per = (bm + bn) // 8 // (wm * wn)  # pieces per wave and K tile; tiles t+1 .. t+STAGES-2 stay in flight at the wait for tile t
        return {"loop": {2 * per: (per, 0), per: (per, 0)}, "prologue": []}
    return None


def _functions(text):
    cur, name, out = [], None, []
    for i, raw in enumerate(text.split("\n")):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            if name:
                out.append((name, cur))
            name, cur = m.group(1), []
            continue
        if name is not None:
            if raw.startswith(".Lfunc_end"):
                out.append((name, cur))
                name, cur = None, []
                continue
            cur.append((i + 1, raw))
    if name:
        out.append((name, cur))
    return out


def _blocks(lines):
    """Basic blocks of one function: [{label, loop header it belongs to, instrs [(ln, op, text, in_asm)], succ labels, falls through}]"""
    blocks, cur, in_asm = [], {"label": "entry", "loop": None, "ins": [], "succ": [], "fall": True}, False
    for ln, raw in lines:
        l = raw.strip()
        if not l:
            continue
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        m2 = re.match(r"^; %bb\.(\d+):", l)
        if m or m2:
            blocks.append(cur)
            lab = m.group(1) if m else "%bb." + m2.group(1)
            lh = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            hdr = re.search(r"Loop Header", l)
            loop = "." + "L" + lh.group(1) if lh else (lab if hdr else None)
            cur = {"label": lab, "loop": loop, "ins": [], "succ": [], "fall": True}
            continue
        if l[0] in ";.":
            continue
        op = l.split()[0]
        cur["ins"].append((ln, op, l, in_asm))
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["succ"].append(l.split()[1])
            if op == "s_branch":
                cur["fall"] = False
            if op == "s_branch":
                blocks.append(cur)
                cur = {"label": "_after_%d" % ln, "loop": cur["loop"], "ins": [], "succ": [], "fall": True}
        if op in ("s_endpgm", "s_setpc_b64"):
            cur["fall"] = False
    blocks.append(cur)
    return blocks


def lint_vmcnt(text):
    """[(kernel, line, message)] for every violation of the rules (2) above."""
    hits = []
    for kernel, lines in _functions(text):
        spec = vmcnt_spec(kernel)
        blocks = _blocks(lines)
        counted = [(b, ins) for b in blocks for ins in b["ins"] if ins[3] and ins[1] == "s_waitcnt" and re.search(r"vmcnt\(([1-9]\d*)\)", ins[2])]
        if not counted:
            continue
        if spec is None:
            hits.append((kernel, counted[0][1][0], "counted asm vmcnt wait in a kernel without a VMCNT_SPECS entry: %s" % counted[0][1][2]))
            continue
        index = {b["label"]: k for k, b in enumerate(blocks)}
        loops = sorted({b["loop"] for b, _ in counted if b["loop"]})
        seen_loop_imms = set()
        for hdr in loops:
            body = [k for k, b in enumerate(blocks) if b["loop"] == hdr]
            inside = set(body)
            for k in body:  # rule a
                for ln, op, l, _ in blocks[k]["ins"]:
                    if _vm_op(op) and op != "global_load_lds_dwordx4":
                        hits.append((kernel, ln, "vector-memory instruction inside a loop with a counted LDS-DMA wait: `%s`" % l))
            start = index[hdr]
            # rule b: over the paths once around the loop (the source guards its pieces with conditions the CFG cannot relate -- `t + 2 < nk` implies
            # `t + 1 < nk` --, so the path that issues the MOST pieces through a wait is the steady-state one; rule a and the static count below exclude
            # a duplicated piece on another path)
            paths, stack, best = 0, [(start, 0, [], False)], {}  # (block, pieces so far, [(imm, pieces before, line)], left the header already)
            while stack:
                k, n, waits, moved = stack.pop()
                if moved and k == start:
                    paths += 1
                    for imm, before, ln in waits:
                        if imm not in best or n > best[imm][0]:
                            best[imm] = (n, before, ln)
                    continue
                if k not in inside or paths > 65536:
                    continue
                b = blocks[k]
                n2, waits2 = n, list(waits)
                for ln, op, l, asm in b["ins"]:
                    if op == "global_load_lds_dwordx4":
                        n2 += 1
                    elif asm and op == "s_waitcnt":
                        mm = re.search(r"vmcnt\((\d+)\)", l)
                        if mm and int(mm.group(1)) > 0:
                            waits2.append((int(mm.group(1)), n2, ln))
                nxt = [index[s] for s in b["succ"] if s in index]
                if b["fall"] and k + 1 < len(blocks):
                    nxt.append(k + 1)
                for t in nxt:
                    stack.append((t, n2, waits2, True))
            for imm, (n, before, ln) in sorted(best.items()):
                seen_loop_imms.add(imm)
                want = spec["loop"].get(imm)
                if want is None:
                    hits.append((kernel, ln, "counted wait vmcnt(%d) is not one the source places in this loop (%s)" % (imm, sorted(spec["loop"]))))
                elif (n, before) != want:
                    hits.append((kernel, ln, "vmcnt(%d): the loop iteration issues %d LDS-DMA pieces, %d of them before the wait; the source counted %d / %d"
                                 % (imm, n, before, want[0], want[1])))
            static = sum(1 for k in body for ins in blocks[k]["ins"] if ins[1] == "global_load_lds_dwordx4")
            per_iter = max(v[0] for v in best.values()) if best else 0  # (a loop unswitched on a launch-uniform flag holds one variant's pieces only)
            if best and static != per_iter:
                hits.append((kernel, blocks[start]["ins"][0][0] if blocks[start]["ins"] else 0,
                             "the loop holds %d LDS-DMA instructions, the source issues %d per iteration (a duplicated or hoisted piece?)" % (static, per_iter)))
            if paths == 0:
                hits.append((kernel, blocks[start]["ins"][0][0] if blocks[start]["ins"] else 0, "no path around the loop with the counted wait was found"))
        for imm in spec["loop"]:
            if imm not in seen_loop_imms:
                hits.append((kernel, 0, "the source's loop wait vmcnt(%d) was not found on any path around a loop" % imm))
        # rule c: the prologue waits.  The code in front of a main loop is scanned in LAYOUT order (its branches re-test launch-uniform conditions --
        # K of one tile? -- that a path enumeration cannot relate to each other; the source issues the pieces and the wait in one straight sequence, and
        # that is what the text must show): pieces from the region's start up to each counted wait; any other vector-memory instruction between the
        # first piece and the region's last counted wait is a hit.  A region = the blocks between two loops (a kernel with two tile instances has two).
        want = list(spec["prologue"])
        regions, cur = [], []
        for b in blocks:
            if b["loop"]:
                if cur:
                    regions.append(cur)
                cur = []
            else:
                cur.append(b)
        if cur:
            regions.append(cur)
        for reg in regions:
            seq = [ins for b in reg for ins in b["ins"]]
            is_counted = lambda ins: ins[3] and ins[1] == "s_waitcnt" and re.search(r"vmcnt\(([1-9]\d*)\)", ins[2])
            last = max([i for i, ins in enumerate(seq) if is_counted(ins)] or [-1])
            n = 0
            for i, (ln, op, l, asm) in enumerate(seq[:last + 1]):
                if op == "global_load_lds_dwordx4":
                    n += 1
                elif _vm_op(op) and n:
                    hits.append((kernel, ln, "vector-memory instruction among the prologue's LDS-DMA pieces: `%s`" % l))
                elif is_counted((ln, op, l, asm)):
                    got = (n, int(re.search(r"vmcnt\((\d+)\)", l).group(1)))
                    if got in want:
                        want.remove(got)
                    else:
                        hits.append((kernel, ln, "prologue: %d pieces then vmcnt(%d); the source counted %s" % (got[0], got[1], spec["prologue"])))
        for pieces, imm in want:
            hits.append((kernel, 0, "the source's prologue wait vmcnt(%d) after %d pieces was not found" % (imm, pieces)))
    return hits


def compiler_version():
    sys.path.insert(0, ROOT)
    from counterfactualworldmodels_amd import build
    return build.hipcc_version()


def compile_isa(src, extra=()):
    sys.path.insert(0, ROOT)
    from counterfactualworldmodels_amd import build
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", *extra, os.path.join(build.CSRC, src), "-o", out]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, res.stderr[-3000:]))
        with open(out) as fh:
            return fh.read()


def lint_source(src):
    """(asm LDS reads, counted vmcnt waits, hits of both rule families) of one source of the library"""
    text = compile_isa(src)
    n_reads = len(re.findall(r";;#ASMSTART\n\s*ds_read_b64_tr_b16", text))
    n_waits = len(re.findall(r";;#ASMSTART\n\s*s_waitcnt vmcnt\([1-9]", text))
    return n_reads, n_waits, lint_isa(text), lint_vmcnt(text)


def record():
    import json

    sys.path.insert(0, ROOT)
    from counterfactualworldmodels_amd import build
    rec = {"hipcc": build.hipcc_version(), "linted": list(SOURCES)}
    with open(build.LINT_RECORD, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("recorded: %s" % rec["hipcc"])


def main():
    rc = 0
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        text = open(sys.argv[1]).read()
        jobs = [(sys.argv[1], (text.count("ds_read_b64_tr_b16"), 0, lint_isa(text), lint_vmcnt(text)))]
    else:
        jobs = [(s, lint_source(s)) for s in SOURCES]
    for src, (n_reads, n_waits, h1, h2) in jobs:
        for k, ln, ins, r, d in h1[:40]:
            print("%s: %s line %d: `%s` (v%d, requested %d lines earlier and not waited for)" % (src, k[:60], ln, ins, r, d))
        for k, ln, msg in h2[:40]:
            print("%s: %s line %d: %s" % (src, k[:60], ln, msg))
        print("%s: %d asm LDS reads, %d counted vmcnt waits, %d + %d hits" % (src, n_reads, n_waits, len(h1), len(h2)))
        rc |= 1 if (h1 or h2) else 0
    if "--record" in sys.argv and rc == 0:
        record()
    return rc


if __name__ == "__main__":
    sys.exit(main())
