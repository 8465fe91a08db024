"""ISA lint for the hand-placed LDS waits of attention_pipe.hip (no GPU needed).

The P V phase requests its V^T fragments with `ds_read_b64_tr_b16` as inline asm and waits for them with hand-counted `s_waitcnt lgkmcnt(N)` asm
statements (attention_device.h lds_read_v_step / lds_wait_v_step).  hipcc believes the asm's output registers are written when the asm statement
issues; it is therefore free to COPY them (v_mov, at a control-flow merge or when it splits a live range) before the wait -- the copy then holds
whatever the register held before, and the kernel is right only as long as the LDS answers faster than the copy comes.  Round 4 hit exactly that: a
branch placed between the reads and their wait made hipcc move 16 fragment registers 85 instructions after the reads; beside another kernel's LDS
traffic (two batch lanes) one wave in ~10^5 read stale fragments (ViT-L/4 batch 8: 5 % of the forwards wrong in one sample).

    python tools/asm_lds_lint.py            # compiles attention_pipe.hip, exits 1 on a hit

Rule checked, per kernel: between a `ds_read_b64_tr_b16 vX` and the first `s_waitcnt lgkmcnt(N)` that covers it (LDS returns in order: a wait for
N leaves the N youngest reads outstanding), no instruction may read vX -- or write it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def lint_isa(text):
    """[(kernel, line number, instruction, register, lines since the read)] for every premature use."""
    hits, kernel, pending, order = [], None, {}, []
    for i, raw in enumerate(text.split("\n")):
        l = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, pending, order = m.group(1), {}, []
            continue
        if not l or l[0] in ";." or kernel is None:
            continue
        op = l.split()[0]
        args = [a for a in l[len(op):].split(",")]
        if op == "ds_read_b64_tr_b16":
            for r in _regs(args[0]):
                pending[r] = i
            order.append(i)
            continue
        if op == "s_waitcnt" and "lgkmcnt" in l:
            n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
            order = order[len(order) - n:] if n > 0 else []
            pending = {r: ln for r, ln in pending.items() if ln in order}
            continue
        if op in ("s_endpgm", "s_setpc_b64"):
            pending, order = {}, []
            continue
        stores = op.startswith("global_store") or op.startswith("ds_write") or op.startswith("buffer_store") or op.startswith("scratch_store")
        for a in (args if stores else args[1:]):
            for r in _regs(a):
                if r in pending:
                    hits.append((kernel, i + 1, l, r, i - pending[r]))
        if not stores and not op.startswith("s_"):  # ... nor may anything else be written there (the late LDS data would land on top of it)
            for r in _regs(args[0]):
                if r in pending:
                    hits.append((kernel, i + 1, l + "   [overwrites]", r, i - pending[r]))
    return hits


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


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        text = compile_isa("attention_pipe.hip")
    hits = lint_isa(text)
    n_reads = text.count("ds_read_b64_tr_b16")
    for k, ln, ins, r, d in hits[:40]:
        print("%s line %d: `%s` reads v%d, requested %d lines earlier and not waited for" % (k[:60], ln, ins, r, d))
    print("%d asm LDS reads, %d premature uses" % (n_reads, len(hits)))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
