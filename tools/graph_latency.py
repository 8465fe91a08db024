"""Small-batch forward: the stream of launches vs the same launches captured once in a hipGraph (torch.cuda.CUDAGraph around the
library call, static input / output tensors) and replayed.   python tools/graph_latency.py [B ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from counterfactualworldmodels_amd import config as C, synthetic as S, vmae  # noqa: E402

cfg = C.CONFIGS["base_8x8patch_2frames_1tube"]
m = vmae.PretrainVisionTransformer(cfg, mode="parity")
m.load_state_dict({k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, 0).items()})
m = m.cuda().eval()
for B in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
    x = torch.from_numpy(S.synthetic_frames(B, cfg, 0)).cuda()
    mask = torch.from_numpy(S.synthetic_masks(B, cfg, 8, 0)).cuda()
    out = torch.empty_like(x)
    run = lambda: m.predict_video(x, mask, n_vis=792, check=False, out_video=out)  # noqa: E731
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    ref = out.clone()

    def timeit(f, n=200):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_stream = timeit(run)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):  # (the stream the warm-up ran on: the library keeps a split-K workspace per stream and allocates it on first use)
        run()
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    t_graph = timeit(g.replay)
    print("B=%d  stream %.3f ms  graph %.3f ms  (%+.1f %%)  outputs identical: %s" % (B, 1e3 * t_stream, 1e3 * t_graph, 100 * (t_graph / t_stream - 1), same), flush=True)
