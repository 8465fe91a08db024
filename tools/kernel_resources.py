"""Register / scratch / LDS use of every kernel of the library, from hipcc's own resource remarks (no GPU needed).

    python tools/kernel_resources.py [source.hip ...] [--scratch-only]

Compiles each source of counterfactualworldmodels_amd/csrc with -Rpass-analysis=kernel-resource-usage (objects go to a temp dir)
and prints one line per kernel.  `--scratch-only` lists the kernels that use scratch memory (spills) and exits 1 if a kernel of
the DEFAULT path is among them (the names in DEFAULT_PATH below).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from counterfactualworldmodels_amd import build  # noqa: E402

# kernels the default configuration of the library launches (substring match on the demangled name)
DEFAULT_PATH = ["gemm8p_kernel<", "gemm_bf16_kernel<2, 128, 128, 2, 4, 2>", "gemm_bf16_kernel<1, 128, 128, 2, 4, 2>",
                "gemm_bf16_kernel<2, 128, 128, 2, 4, 4>", "gemm_bf16_kernel<1, 128, 128, 2, 4, 4>", "gemm_bf16_kernel<2, 64, 128, 1, 4, 4>",
                "gemm_bf16_kernel<1, 64, 128, 1, 4, 4>", "attention_pipe_kernel", "attention_kernel",
                "layernorm_kernel", "patch_gather_kernel", "index_gather_kernel", "unembed3_kernel", "fill_mask_tokens4_kernel", "mask_to_perm_kernel", "shift_prompts_kernel",
                "cross_attn_mfma_kernel", "small_attn_mfma_kernel"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"^void ", "", re.sub(r"\(cwm::.*$|\(.*\)$", "", o)).replace("cwm::", "") for o in out]


def resources(src, extra=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", *extra, "-c",
               os.path.join(build.CSRC, src), "-o", os.path.join(td, "o.o")]
        res = subprocess.run(cmd, capture_output=True, text=True)
        err = res.stderr
        if res.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, err[-3000:]))
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = n
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    scratch_only = "--scratch-only" in sys.argv
    bad = 0
    for src in (args or build.SOURCES):
        for r in resources(src):
            if scratch_only and not r.get("scratch"):
                continue
            default = any(d in r["name"] for d in DEFAULT_PATH)
            if r.get("scratch") and default:
                bad += 1
            print("%-18s %-70s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d%s" % (src, r["name"][:70], r.get("vgpr", -1), r.get("agpr", 0), r.get("sgpr", -1),
                                                                                   r.get("scratch", 0), r.get("occ", -1), "  [default path]" if default else ""))
    if scratch_only and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
