"""Share of a file's non-blank lines that also appear (whitespace-normalised) in a reference file.
    python tools/line_overlap.py counterfactualworldmodels_amd/prediction.py /root/reference/cwm/models/prediction.py [more refs...]
Development aid (build container only: the reference does not exist on the GPU box)."""
import re
import sys


def norm(line):
    return re.sub(r"\s+", "", line.split("#")[0]) if not line.strip().startswith("#") else ""


def lines(path):
    with open(path) as f:
        return [n for n in (norm(l) for l in f) if len(n) > 3]


mine = lines(sys.argv[1])
ref = set()
for r in sys.argv[2:]:
    ref.update(lines(r))
hits = [l for l in mine if l in ref]
print("%d of %d lines (%.1f %%) appear in the reference" % (len(hits), len(mine), 100.0 * len(hits) / max(len(mine), 1)))
if "-v" in sys.argv or True:
    for l in hits:
        print("   ", l[:110])
