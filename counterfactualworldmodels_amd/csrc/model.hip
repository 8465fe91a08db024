// Model handle, weight packing, workspace and forward orchestration behind the C ABI (include/cwm_hip.h).
// Restates the control flow of `PretrainVisionTransformer.forward` (cwm/models/VideoMAE/vmae.py:539-560),
// `PretrainVisionTransformerEncoder.forward_features` (:152-173), `...Decoder.forward` (:246-255) and
// `Block.forward` (cwm/models/VideoMAE/utils.py:146-153) as a fixed sequence of HIP kernel launches
// on the caller's stream.
#include <stddef.h>

#include "engine.h"

using namespace cwm;

struct cwm_model {
    Engine eng;
    cwm_config cfg;
    int Nt = 0, n_per_frame = 0, patch_k = 0, patch_kpad = 0, out_dim = 0;
    std::vector<BlockW> enc, dec;
    LinearW patch, e2d, head;
    float *enc_norm_g = nullptr, *enc_norm_b = nullptr, *dec_norm_g = nullptr, *dec_norm_b = nullptr;
    float* mask_token = nullptr;
    float *pos_enc = nullptr, *pos_dec = nullptr;  // [Nt][De], [Nt][Dd] sinusoid tables
    // workspace (grown on demand)
    int ws_batch = 0, ws_nvis = 0;
    int *perm = nullptr, *rank = nullptr, *err = nullptr;
    bf16* patches = nullptr;
    float *x_enc = nullptr, *x_dec = nullptr;
    StreamBuffers sb;
    // batch lanes (cwm_model_set_lanes): a batch whose halves keep >= kMinLaneRows encoder rows runs as two half batches, the first on the
    // caller's stream and the second on `lane_stream`, joined by events before cwm_forward returns control of the stream
    int lanes = 2;
    static constexpr int kMaxLanes = 4;
    hipStream_t lane_stream[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
    ~cwm_model() {
        for (auto s : lane_stream)
            if (s) (void)hipStreamDestroy(s);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (auto e : ev_join)
            if (e) (void)hipEventDestroy(e);
    }
};

// per-lane view of the workspace: every buffer is batch-major, so the lane that starts at batch element b0 owns the slice
// behind the capacity of b0 elements
struct LaneWs {
    int *perm, *rank, *err;
    bf16* patches;
    float *x_enc, *x_dec;
    StreamBuffers sb;
};

namespace {

int ensure_workspace(cwm_model* m, int B, int n_vis) {
    if (B <= m->ws_batch && n_vis <= m->ws_nvis && m->ws_batch > 0) return 0;
    Engine& E = m->eng;
    if (int rc = E.free_workspace()) return rc;
    const cwm_config& c = m->cfg;
    const int Bc = std::max(B, m->ws_batch), Nv = std::max(n_vis, m->ws_nvis), Nt = m->Nt;
    const size_t rows_e = (size_t)Bc * Nv, rows_d = (size_t)Bc * Nt;
    int rc;
    if ((rc = E.ws(&m->perm, rows_d)) || (rc = E.ws(&m->rank, rows_d)) || (rc = E.ws(&m->err, (size_t)Bc + 4))) return rc;
    if ((rc = E.ws(&m->patches, 2 * rows_e * m->patch_kpad))) return rc;
    if ((rc = E.ws(&m->x_enc, rows_e * c.enc_dim)) || (rc = E.ws(&m->x_dec, rows_d * c.dec_dim))) return rc;
    const size_t act = std::max(rows_e * c.enc_dim, rows_d * c.dec_dim);
    if ((rc = E.ws(&m->sb.hbuf, 2 * act)) || (rc = E.ws(&m->sb.gbuf, 2 * act * c.mlp_ratio)) || (rc = E.ws(&m->sb.qbuf, 2 * act)) ||
        (rc = E.ws(&m->sb.kbuf, 2 * act)))
        return rc;
    if ((rc = E.ws(&m->sb.vbuf, 2 * act))) return rc;
    m->ws_batch = Bc;
    m->ws_nvis = Nv;
    // (the zero fills above ran on the null stream; the lane streams are non-blocking and would not wait for them)
    CWM_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

LaneWs lane_ws(const cwm_model* m, int lane, int b0) {
    const cwm_config& c = m->cfg;
    const size_t rows_e = (size_t)b0 * m->ws_nvis, rows_d = (size_t)b0 * m->Nt;
    const size_t act = std::max(rows_e * c.enc_dim, rows_d * c.dec_dim);
    LaneWs w;
    w.perm = m->perm + rows_d;
    w.rank = m->rank + rows_d;
    w.err = m->err + b0;  // one word per batch row (index_gather_kernel writes every row's, every call: no memset)
    (void)lane;
    w.patches = m->patches + 2 * rows_e * m->patch_kpad;
    w.x_enc = m->x_enc + rows_e * c.enc_dim;
    w.x_dec = m->x_dec + rows_d * c.dec_dim;
    w.sb = m->sb;
    w.sb.hbuf += 2 * act;
    w.sb.gbuf += 2 * act * c.mlp_ratio;
    w.sb.qbuf += 2 * act;
    w.sb.kbuf += 2 * act;
    w.sb.vbuf += 2 * act;
    return w;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int cwm_model_create(const cwm_config* cfg, cwm_model** out) {
    CWM_REQUIRE(cfg && out, "cwm_model_create: null argument");
    const cwm_config& c = *cfg;
    CWM_REQUIRE(c.patch > 0 && c.img_h % c.patch == 0 && c.img_w % c.patch == 0, "image size (%d,%d) must be divisible by patch size %d",
                c.img_h, c.img_w, c.patch);
    CWM_REQUIRE(c.patch % 4 == 0 && c.img_w % 4 == 0, "patch size must be a multiple of 4");
    CWM_REQUIRE(c.enc_heads > 0 && c.dec_heads > 0 && c.enc_dim == 64 * c.enc_heads && c.dec_dim == 64 * c.dec_heads,
                "this build supports head_dim 64 only (enc %d/%d, dec %d/%d)", c.enc_dim, c.enc_heads, c.dec_dim, c.dec_heads);
    CWM_REQUIRE(c.enc_dim <= 1024 && c.dec_dim <= 1024 && c.enc_dim % 64 == 0 && c.dec_dim % 64 == 0, "embed dims must be multiples of 64, <= 1024");
    CWM_REQUIRE(c.in_chans == 3 && c.num_frames >= 1 && c.mlp_ratio >= 1 && c.enc_depth >= 1 && c.dec_depth >= 1, "unsupported config");
    cwm_model* m = new cwm_model();
    m->cfg = c;
    Engine& E = m->eng;
    E.ln_eps = c.ln_eps;
    CWM_HIP_CHECK(hipGetDevice(&E.device));
    m->n_per_frame = (c.img_h / c.patch) * (c.img_w / c.patch);
    m->Nt = m->n_per_frame * c.num_frames;
    m->patch_k = c.in_chans * c.patch * c.patch;
    m->patch_kpad = round_up(m->patch_k, 64);
    m->out_dim = c.in_chans * c.patch * c.patch;
    int rc = 0;
    m->enc.resize(c.enc_depth);
    m->dec.resize(c.dec_depth);
    do {
        if ((rc = E.make_linear(m->patch, c.enc_dim, m->patch_k, true))) break;
        E.add_matrix_slot("encoder.patch_embed.proj.weight", &m->patch, {c.enc_dim, c.in_chans, 1, c.patch, c.patch});
        E.add_vec_slot("encoder.patch_embed.proj.bias", m->patch.bias, {c.enc_dim});
        for (int i = 0; i < c.enc_depth && !rc; ++i)
            rc = E.make_block(m->enc[i], "encoder.blocks." + std::to_string(i) + ".", c.enc_dim, c.mlp_ratio * c.enc_dim);
        if (rc) break;
        if ((rc = E.make_vec(&m->enc_norm_g, c.enc_dim)) || (rc = E.make_vec(&m->enc_norm_b, c.enc_dim))) break;
        E.add_vec_slot("encoder.norm.weight", m->enc_norm_g, {c.enc_dim});
        E.add_vec_slot("encoder.norm.bias", m->enc_norm_b, {c.enc_dim});
        if ((rc = E.make_linear(m->e2d, c.dec_dim, c.enc_dim, false))) break;
        E.add_matrix_slot("encoder_to_decoder.weight", &m->e2d, {c.dec_dim, c.enc_dim});
        if ((rc = E.make_vec(&m->mask_token, c.dec_dim))) break;
        E.add_vec_slot("mask_token", m->mask_token, {1, 1, c.dec_dim});
        for (int i = 0; i < c.dec_depth && !rc; ++i)
            rc = E.make_block(m->dec[i], "decoder.blocks." + std::to_string(i) + ".", c.dec_dim, c.mlp_ratio * c.dec_dim);
        if (rc) break;
        if ((rc = E.make_vec(&m->dec_norm_g, c.dec_dim)) || (rc = E.make_vec(&m->dec_norm_b, c.dec_dim))) break;
        E.add_vec_slot("decoder.norm.weight", m->dec_norm_g, {c.dec_dim});
        E.add_vec_slot("decoder.norm.bias", m->dec_norm_b, {c.dec_dim});
        if ((rc = E.make_linear(m->head, m->out_dim, c.dec_dim, true))) break;
        E.add_matrix_slot("decoder.head.weight", &m->head, {m->out_dim, c.dec_dim});
        E.add_vec_slot("decoder.head.bias", m->head.bias, {m->out_dim});
        if ((rc = E.make_sinusoid(&m->pos_enc, m->Nt, c.enc_dim))) break;  // vmae.py:75
        if ((rc = E.make_sinusoid(&m->pos_dec, m->Nt, c.dec_dim))) break;  // vmae.py:366
    } while (0);
    if (rc) {
        cwm_model_destroy(m);
        return rc;
    }
    *out = m;
    return CWM_OK;
}

extern "C" void cwm_model_destroy(cwm_model* m) { delete m; }

extern "C" int cwm_model_load_weight(cwm_model* m, const char* key, const float* data, int on_device, const int64_t* shape, int ndim) {
    CWM_REQUIRE(m, "cwm_model_load_weight: null model");
    return m->eng.load_weight(key, data, on_device, shape, ndim);
}

extern "C" int cwm_model_missing_weights(cwm_model* m, char* buf, int buflen) { return m->eng.missing_weights(buf, buflen); }

// (Rounds 2-3 carried a form with every LayerNorm folded into the GEMMs around it -- parity-tested, measured slower: the 4 bytes / element
// a producer GEMM then adds to its store-bound epilogue cost more than the LayerNorm launch they replace, DESIGN.md section 4.6 -- never the
// default and removed in round 4.)

// One lane: batch elements [b0, b0 + B) of the call, on stream s, in the workspace slice w.
// Stages [stage_lo, stage_hi) of the lane's launch sequence: 0 = mask -> permutation, patch gather + embed; 1 .. Le = encoder blocks;
// Le + 1 = encoder norm + encoder_to_decoder + mask tokens; Le + 2 .. Le + 1 + Ld = decoder blocks; Le + Ld + 2 = norm + head + un-embed.
// (cwm_forward issues stage by stage over all lanes, so that after a host synchronisation every lane's queue starts filling at once
// instead of lane 1 waiting behind lane 0's ~136 launches.)
static int forward_lane(cwm_model* m, const cwm_forward_args* a, int b0, int B, LaneWs w, hipStream_t s, int stage_lo, int stage_hi) {
    const cwm_config& c = m->cfg;
    const int Nt = m->Nt, Nv = a->n_vis, Nm = Nt - Nv;
    const int Nret = Nm > 0 ? Nm : Nt;
    Engine& E = m->eng;
    const int planes = a->mode == CWM_MODE_PARITY ? 2 : 1;
    const float* x_in = a->x_dev + (int64_t)b0 * a->x_stride_b;
    const uint8_t* mask_in = a->mask_dev + (size_t)b0 * Nt;
    float* y_tokens = a->y_tokens_dev + (size_t)b0 * Nret * m->out_dim;
    int rc;
    auto in_range = [&](int st) { return st >= stage_lo && st < stage_hi; };
    const int st_e2d = c.enc_depth + 1, st_head = c.enc_depth + c.dec_depth + 2;
    GemmParams g;
    LayerNormParams ln;

    if (in_range(0)) {
    // a1-a3: mask -> permutation (+ its inverse for the un-embed, + the per-row check of the visible count), frame load (+normalise) and
    // tubelet patch gather of the visible tokens in ONE launch (elementwise.hip index_gather_kernel); then the patch-embed GEMM with bias
    // and positional-table add in the epilogue
    PatchGatherParams pg;
    memset(&pg, 0, sizeof(pg));
    pg.x = x_in; pg.sb = a->x_stride_b; pg.sc = a->x_stride_c; pg.st = a->x_stride_t; pg.normalize = a->normalize;
    pg.C = c.in_chans; pg.H = c.img_h; pg.W = c.img_w; pg.P = c.patch; pg.perm = w.perm; pg.Nt = Nt; pg.n_rows = Nv; pg.B = B;
    pg.out = w.patches; pg.out_plane = (int64_t)B * Nv * m->patch_kpad; pg.ld = m->patch_kpad;
    if (E.tune.index_fused) {
        if ((rc = E.run_index_gather(pg, mask_in, Nv, w.perm, a->y_video_dev ? w.rank : nullptr, w.err, planes, s))) return rc;
    } else {  // the four launches of rounds 1-4 (A/B and the bitwise cross-check of the fused kernel)
        CWM_HIP_CHECK(hipMemsetAsync(w.err, 0, (size_t)B * sizeof(int), s));
        if ((rc = launch_mask_to_perm(mask_in, B, Nt, Nv, w.perm, w.err, s))) return rc;
        if ((rc = E.run_patch_gather(pg, planes, s))) return rc;
        if (a->y_video_dev && (rc = launch_perm_to_rank(w.perm, w.rank, B, Nt, s))) return rc;
    }
    }

    StreamBuffers sb_enc = w.sb, sb_dec = w.sb;
    if (in_range(0)) {
    g = gemm_base(w.patches, m->patch_kpad, m->patch, B * Nv, planes);
    g.epi = EPI_F32; g.C = w.x_enc; g.ldc = c.enc_dim;
    g.resid = m->pos_enc; g.ldr = c.enc_dim; g.resid_rowmap = w.perm; g.rows_in = Nv; g.rows_out = Nv; g.map_stride = Nt;
    if ((rc = E.run_gemm(g, planes, s))) return rc;
    }

    // a4-a6: encoder blocks over the visible tokens
    for (int i = 0; i < c.enc_depth; ++i)
        if (in_range(1 + i) && (rc = E.run_block(m->enc[i], w.x_enc, B, Nv, c.enc_dim, c.enc_heads, planes, sb_enc, s))) return rc;

    // a7: encoder.norm, encoder_to_decoder (no bias); a8: + pos[vis] written straight into x_full rows [0,Nv)
    if (in_range(st_e2d)) {
    memset(&ln, 0, sizeof(ln));
    ln.x = w.x_enc; ln.ldx = c.enc_dim; ln.gamma = m->enc_norm_g; ln.beta = m->enc_norm_b; ln.eps = c.ln_eps; ln.D = c.enc_dim;
    ln.rows = B * Nv; ln.out = w.sb.hbuf; ln.out_plane = (int64_t)B * Nv * c.enc_dim; ln.ldo = c.enc_dim;
    if ((rc = E.run_layernorm(ln, planes, s))) return rc;
    g = gemm_base(w.sb.hbuf, c.enc_dim, m->e2d, B * Nv, planes);
    g.epi = EPI_F32; g.C = w.x_dec; g.ldc = c.dec_dim;
    g.resid = m->pos_dec; g.ldr = c.dec_dim; g.resid_rowmap = w.perm; g.rows_in = Nv; g.rows_out = Nt; g.map_stride = Nt;
    if ((rc = E.run_gemm(g, planes, s))) return rc;
    if (Nm > 0 && (rc = E.run_fill_mask_tokens(w.x_dec, m->mask_token, m->pos_dec, w.perm, B, Nt, Nv, c.dec_dim, s))) return rc;
    }

    // a9: decoder blocks over the full token set, norm + head on the last Nm tokens
    // (the last block only has to produce the Nm rows the head reads: option "prune_last_block" = 0 runs it in full)
    const bool pruned = Nm > 0 && E.tune.prune_last_block;
    for (int i = 0; i < c.dec_depth; ++i) {
        const int keep = (i == c.dec_depth - 1 && pruned) ? Nm : 0;
        if (in_range(st_e2d + 1 + i) && (rc = E.run_block(m->dec[i], w.x_dec, B, Nt, c.dec_dim, c.dec_heads, planes, sb_dec, s, keep))) return rc;
    }
    if (!in_range(st_head)) return CWM_OK;
    memset(&ln, 0, sizeof(ln));
    ln.x = w.x_dec; ln.ldx = c.dec_dim; ln.gamma = m->dec_norm_g; ln.beta = m->dec_norm_b; ln.eps = c.ln_eps; ln.D = c.dec_dim;
    ln.rows = B * Nret; ln.rows_out_per_b = Nret; ln.rows_in_per_b = Nt; ln.in_offset = Nt - Nret;
    ln.out = w.sb.hbuf; ln.out_plane = (int64_t)B * Nret * c.dec_dim; ln.ldo = c.dec_dim;
    if ((rc = E.run_layernorm(ln, planes, s))) return rc;
    g = gemm_base(w.sb.hbuf, c.dec_dim, m->head, B * Nret, planes);
    g.epi = EPI_F32; g.C = y_tokens; g.ldc = m->out_dim;
    if ((rc = E.run_gemm(g, planes, s))) return rc;

    // a11: patch un-embed scatter
    if (a->y_video_dev) {
        const float* xr = a->xraw_dev ? a->xraw_dev + (int64_t)b0 * a->x_stride_b : x_in;
        UnembedParams u;  // (w.rank: written by the index prologue of stage 0)
        memset(&u, 0, sizeof(u));
        u.y = y_tokens; u.x = xr; u.sb = a->x_stride_b; u.sc = a->x_stride_c; u.st = a->x_stride_t;
        u.mask = mask_in; u.rank = w.rank; u.B = B; u.T = c.num_frames; u.C = c.in_chans; u.H = c.img_h; u.W = c.img_w;
        u.P = c.patch; u.n_vis = Nv; u.Nm = Nm;
        u.out = a->y_video_dev + (size_t)b0 * c.num_frames * c.in_chans * c.img_h * c.img_w;
        if ((rc = E.run_unembed(u, s))) return rc;
    }
    return CWM_OK;
}

extern "C" int cwm_forward(cwm_model* m, const cwm_forward_args* a_in) {
    CWM_REQUIRE(m && a_in, "cwm_forward: null argument");
    // the caller's struct may end before fields a later version appends: copy what it has, the rest stays zero (= not requested).  The upper bound
    // catches a caller built against the 0.5 header (no struct_size: the low half of its x_dev pointer lands here).
    CWM_REQUIRE(a_in->struct_size >= offsetof(cwm_forward_args, stream) + sizeof(void*) && a_in->struct_size <= 4096,
                "cwm_forward: args->struct_size = %u is not a cwm_forward_args (set it to sizeof(cwm_forward_args); callers built against the 0.5 header must be rebuilt)",
                a_in->struct_size);
    cwm_forward_args a_copy;
    memset(&a_copy, 0, sizeof(a_copy));
    memcpy(&a_copy, a_in, std::min<size_t>(a_in->struct_size, sizeof(a_copy)));
    const cwm_forward_args* a = &a_copy;
    if (int rc = cwm_require_device(m->eng.device, "cwm_forward")) return rc;
    CWM_REQUIRE(a->x_dev && a->mask_dev && a->y_tokens_dev, "cwm_forward: x_dev, mask_dev and y_tokens_dev are required");
    CWM_REQUIRE(a->mode == CWM_MODE_FAST || a->mode == CWM_MODE_PARITY, "cwm_forward: bad mode %d", a->mode);
    const cwm_config& c = m->cfg;
    const int B = a->batch, Nt = m->Nt, Nv = a->n_vis, Nm = Nt - Nv;
    CWM_REQUIRE(B > 0, "cwm_forward: batch must be positive");
    CWM_REQUIRE(Nv > 0 && Nm >= 0, "cwm_forward: need 0 < n_vis (%d) <= num tokens (%d)", Nv, Nt);
    // nothing masked: the reference decoder then returns head(norm(x)) for ALL tokens (vmae.py:250-253), and its wrapper cannot
    // compose a video from that (prediction.py:252-254 would assign Nt rows to an empty selection)
    CWM_REQUIRE(Nm > 0 || !a->y_video_dev, "cwm_forward: no token is masked, there is no predicted patch to un-embed");
    CWM_REQUIRE(!a->y_video_dev || a->xraw_dev || a->normalize, "cwm_forward: y_video_dev needs the raw frames (xraw_dev) when normalize=0");
    {
        char miss[256];
        const int nmiss = m->eng.missing_weights(miss, sizeof(miss));
        CWM_REQUIRE(nmiss == 0, "cwm_forward: %d state-dict tensors not loaded (first: %s)", nmiss, miss);
    }
    if (int rc = ensure_workspace(m, B, Nv)) return rc;
    hipStream_t s = (hipStream_t)a->stream;

    // Batch lanes: between two dependent kernels the queue idles ~6 us (x 136 kernels = 5 % of a batch-32 step) and every kernel ends in a
    // partially filled round of workgroups; further, independent slices of the batch on other queues fill both.
    const int min_rows = m->eng.tune.min_lane_rows > 0 ? m->eng.tune.min_lane_rows : kMinLaneRows;
    int n_lanes = 1;
    while (n_lanes < m->lanes && n_lanes < B && (int64_t)(B / (n_lanes + 1)) * Nv >= min_rows) ++n_lanes;
    const bool two = n_lanes >= 2;
    int first[cwm_model::kMaxLanes + 1];
    for (int l = 0; l <= n_lanes; ++l) first[l] = (int)(((int64_t)B * l + n_lanes - 1) / n_lanes);  // lane l owns batch elements [first[l], first[l+1])
    if (two) {
        if (!m->ev_fork) CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        CWM_HIP_CHECK(hipEventRecord(m->ev_fork, s));  // inputs written on the caller's stream are complete for the other lanes
        for (int l = 1; l < n_lanes; ++l) {
            if (!m->lane_stream[l - 1]) {
                CWM_HIP_CHECK(hipStreamCreateWithFlags(&m->lane_stream[l - 1], hipStreamNonBlocking));
                CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join[l - 1], hipEventDisableTiming));
            }
            CWM_HIP_CHECK(hipStreamWaitEvent(m->lane_stream[l - 1], m->ev_fork, 0));
        }
    }
    m->eng.overlapped = two;
    int rc = 0;
    // launch order: stage by stage (one transformer block at a time) over all lanes, so that every lane's queue starts filling at once
    const int n_stages = c.enc_depth + c.dec_depth + 3;
    for (int st = 0; st < n_stages && !rc; ++st)
        for (int l = 0; l < n_lanes && !rc; ++l)
            rc = forward_lane(m, a, first[l], first[l + 1] - first[l], lane_ws(m, l, first[l]), l == 0 ? s : m->lane_stream[l - 1], st, st + 1);
    m->eng.overlapped = 0;
    // join even after a failed launch: the caller's stream must not run ahead of work already queued on a lane
    for (int l = 1; l < n_lanes; ++l) {
        CWM_HIP_CHECK(hipEventRecord(m->ev_join[l - 1], m->lane_stream[l - 1]));
        CWM_HIP_CHECK(hipStreamWaitEvent(s, m->ev_join[l - 1], 0));
    }
    if (rc) return rc;

    if (a->check) {
        std::vector<int> herr((size_t)B, 0);
        CWM_HIP_CHECK(hipMemcpyAsync(herr.data(), m->err, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
        CWM_HIP_CHECK(hipStreamSynchronize(s));
        if (std::any_of(herr.begin(), herr.end(), [](int e) { return e != 0; })) {
            cwm_set_error("mask rows do not all have n_vis=%d visible tokens (shape '[%d, -1, %d]' is invalid for the gathered input)", Nv, B,
                          c.enc_dim);
            return CWM_ERR_MASK;
        }
    }
    return CWM_OK;
}

extern "C" int cwm_model_set_lanes(cwm_model* m, int lanes) {
    CWM_REQUIRE(m && lanes >= 1 && lanes <= cwm_model::kMaxLanes, "cwm_model_set_lanes: lanes must be 1 .. %d", cwm_model::kMaxLanes);
    m->lanes = lanes;
    return CWM_OK;
}

extern "C" int cwm_model_set_option(cwm_model* m, const char* key, int value) {
    CWM_REQUIRE(m && key, "cwm_model_set_option: null argument");
    const int rc = tuning_set_production(m->eng.tune, key, value);
    CWM_REQUIRE(rc != -2, "cwm_model_set_option: gemm_debug bits 1, 2 and 8 are timing-only ablations (wrong outputs): development library only (cwm_debug_set)");
    CWM_REQUIRE(rc == 0, "cwm_model_set_option: unknown option %s", key);
    return CWM_OK;
}

extern "C" int cwm_timing_enable(cwm_model* m, int kclass, int enable) {
    CWM_REQUIRE(m, "cwm_timing_enable: null model");
    return m->eng.timing_enable(kclass, enable);
}

extern "C" int cwm_timing_collect(cwm_model* m, int kclass, cwm_kernel_stats* out) {
    CWM_REQUIRE(m, "cwm_timing_collect: null model");
    return m->eng.timing_collect(kclass, out);
}

// ---------------------------------------------------------------------------------------------
// stand-alone kernel entry points (tests).  They allocate scratch per call: not for hot loops.
// ---------------------------------------------------------------------------------------------
namespace {
struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    T* get(size_t count, bool zero = false) {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(T) + 16) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        if (zero) (void)hipMemset(p, 0, count * sizeof(T));
        return (T*)p;
    }
};

// fp32 [rows][K] -> GEMM A-operand layout of the given mode (common.h a_pos), K zero-padded to Kpad
template <int PLANES>
__global__ void pad_split_rows_kernel(const float* src, int rows, int K, bf16* out, int Kpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * Kpad) return;
    const int r = (int)(i / Kpad), k = (int)(i - (int64_t)r * Kpad);
    const float v = k < K ? src[(size_t)r * K + k] : 0.f;
    bf16 h, l;
    split_bf16(v, h, l);
    bf16* d = out + a_pos<PLANES>(r, Kpad, k);
    d[0] = h;
    if constexpr (PLANES == 2) d[kLoOffset] = l;
}

// qkv [B,N,3,H,64] fp32 -> Q (scaled), K, V [B*H,N,64]
__global__ void qkv_scatter_kernel(const float* qkv, int B, int N, int H, float scale, bf16* q, bf16* k, bf16* v_out, int64_t qk_plane) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = H * 64;
    if (i >= (int64_t)B * N * 3 * D) return;
    const int c = (int)(i % (3 * D));
    const int64_t row = i / (3 * D);
    const int b = (int)(row / N), n = (int)(row - (int64_t)b * N);
    const int which = c / D, cc = c - which * D, h = cc / 64, d = cc - h * 64;
    float v = qkv[i];
    if (which == 0) v *= scale;
    bf16 hi, lo;
    split_bf16(v, hi, lo);
    bf16* dst = which == 0 ? q : which == 1 ? k : v_out;
    const size_t o = ((size_t)(b * H + h) * N + n) * 64 + d;
    dst[o] = hi;
    dst[o + qk_plane] = lo;
}

// GEMM A-operand layout (row length ld_in, a multiple of 32) -> fp32 [rows][ld]
template <int PLANES>
__global__ void merge_planes_kernel(const bf16* in, float* out, int rows, int ld, int ld_in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * ld) return;
    const int r = (int)(i / ld), c = (int)(i - (int64_t)r * ld);
    const bf16* s = in + a_pos<PLANES>(r, ld_in, c);
    out[i] = (float)s[0] + (PLANES == 2 ? (float)s[kLoOffset] : 0.f);
}
}  // namespace

extern "C" int cwm_split_bf16(const float* x_dev, int64_t n, void* hi_dev, void* lo_dev, void* stream) {
    CWM_REQUIRE(x_dev && hi_dev && n >= 0, "cwm_split_bf16: bad argument");
    return launch_split_bf16(x_dev, n, (bf16*)hi_dev, (bf16*)lo_dev, (hipStream_t)stream);
}

extern "C" int cwm_linear(const float* a_dev, const float* w_dev, const float* bias_dev, const float* resid_dev, float* c_dev, int M,
                          int N, int K, int gelu, int mode, void* stream) {
    CWM_REQUIRE(a_dev && w_dev && c_dev && M > 0 && N > 0 && K > 0, "cwm_linear: bad argument");
    CWM_REQUIRE(mode == CWM_MODE_FAST || mode == CWM_MODE_PARITY, "cwm_linear: bad mode");
    hipStream_t s = (hipStream_t)stream;
    const int planes = mode == CWM_MODE_PARITY ? 2 : 1;
    const int Kp = round_up(K, 64), Np = round_up(N, 256);
    Scratch sc;
    bf16* A = sc.get<bf16>((size_t)2 * M * Kp);
    bf16* W = sc.get<bf16>((size_t)2 * Np * Kp);
    float* bias = sc.get<float>(Np, true);
    const int ldg = round_up(N, 32);  // operand-layout rows are whole [32 hi | 32 lo] blocks
    bf16* G = gelu ? sc.get<bf16>((size_t)2 * M * ldg) : nullptr;
    CWM_REQUIRE(A && W && bias && (!gelu || G), "cwm_linear: out of device memory");
    const unsigned gridA = (unsigned)(((int64_t)M * Kp + 255) / 256);
    if (planes == 2)
        hipLaunchKernelGGL(pad_split_rows_kernel<2>, dim3(gridA), dim3(256), 0, s, a_dev, M, K, A, Kp);
    else
        hipLaunchKernelGGL(pad_split_rows_kernel<1>, dim3(gridA), dim3(256), 0, s, a_dev, M, K, A, Kp);
    if (int rc = launch_pack_weight(w_dev, N, K, planes == 1 ? W : nullptr, planes == 2 ? W : nullptr, Np, Kp, s)) return rc;
    if (bias_dev) CWM_HIP_CHECK(hipMemcpyAsync(bias, bias_dev, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, s));
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = Kp; p.W = W;
    p.M = M; p.N = N; p.K = Kp; p.bias = bias;
    if (gelu) {
        p.epi = EPI_BF16_GELU; p.out_hi = G; p.ldo = ldg;
    } else {
        p.epi = EPI_F32; p.C = c_dev; p.ldc = N; p.resid = resid_dev; p.ldr = N;
    }
    p.tune = &thread_tuning();
    if (int rc = launch_gemm(p, planes, s)) return rc;
    if (gelu) {
        const unsigned gridG = (unsigned)(((int64_t)M * N + 255) / 256);
        if (planes == 2)
            hipLaunchKernelGGL(merge_planes_kernel<2>, dim3(gridG), dim3(256), 0, s, G, c_dev, M, N, ldg);
        else
            hipLaunchKernelGGL(merge_planes_kernel<1>, dim3(gridG), dim3(256), 0, s, G, c_dev, M, N, ldg);
    }
    CWM_HIP_CHECK(hipStreamSynchronize(s));
    return CWM_OK;
}

extern "C" int cwm_attention(const float* qkv_dev, float* o_dev, int B, int N, int H, int mode, void* stream) {
    CWM_REQUIRE(qkv_dev && o_dev && B > 0 && N > 0 && H > 0, "cwm_attention: bad argument");
    CWM_REQUIRE(mode == CWM_MODE_FAST || mode == CWM_MODE_PARITY, "cwm_attention: bad mode");
    hipStream_t s = (hipStream_t)stream;
    const int planes = mode == CWM_MODE_PARITY ? 2 : 1;
    const int D = H * 64;
    const int64_t qk_plane = (int64_t)B * N * D;
    Scratch sc;
    bf16* q = sc.get<bf16>(2 * qk_plane);
    bf16* k = sc.get<bf16>(2 * qk_plane);
    bf16* v = sc.get<bf16>(2 * qk_plane);
    bf16* o = sc.get<bf16>(2 * qk_plane);
    CWM_REQUIRE(q && k && v && o, "cwm_attention: out of device memory");
    const int64_t total = (int64_t)B * N * 3 * D;
    hipLaunchKernelGGL(qkv_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv_dev, B, N, H, 0.125f, q, k, v, qk_plane);
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.qk_plane = qk_plane; a.o = o; a.o_plane = qk_plane; a.ldo = D;
    a.n_tok = N; a.heads = H; a.batch = B;
    a.tune = &thread_tuning();
    if (int rc = launch_attention(a, planes, s)) return rc;
    if (planes == 2)
        hipLaunchKernelGGL(merge_planes_kernel<2>, dim3((unsigned)((qk_plane + 255) / 256)), dim3(256), 0, s, o, o_dev, B * N, D, D);
    else
        hipLaunchKernelGGL(merge_planes_kernel<1>, dim3((unsigned)((qk_plane + 255) / 256)), dim3(256), 0, s, o, o_dev, B * N, D, D);
    CWM_HIP_CHECK(hipStreamSynchronize(s));
    return CWM_OK;
}

extern "C" int cwm_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, int rows, int D, float eps,
                             void* stream) {
    CWM_REQUIRE(x_dev && gamma_dev && beta_dev && y_dev && rows > 0, "cwm_layernorm: bad argument");
    hipStream_t s = (hipStream_t)stream;
    Scratch sc;
    bf16* tmp = sc.get<bf16>((size_t)2 * rows * D);
    CWM_REQUIRE(tmp, "cwm_layernorm: out of device memory");
    LayerNormParams ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = x_dev; ln.ldx = D; ln.gamma = gamma_dev; ln.beta = beta_dev; ln.eps = eps; ln.D = D; ln.rows = rows;
    ln.out = tmp; ln.out_plane = (int64_t)rows * D; ln.ldo = D; ln.out_f32 = y_dev;
    if (int rc = launch_layernorm(ln, 2, s)) return rc;
    CWM_HIP_CHECK(hipStreamSynchronize(s));
    return CWM_OK;
}

extern "C" int cwm_mask_to_perm(const uint8_t* mask_dev, int B, int Nt, int n_vis, int32_t* perm_dev, void* stream) {
    CWM_REQUIRE(mask_dev && perm_dev && B > 0 && Nt > 0, "cwm_mask_to_perm: bad argument");
    hipStream_t s = (hipStream_t)stream;
    Scratch sc;
    int* err = sc.get<int>(4, true);
    CWM_REQUIRE(err, "cwm_mask_to_perm: out of device memory");
    if (int rc = launch_mask_to_perm(mask_dev, B, Nt, n_vis, perm_dev, err, s)) return rc;
    int herr = 0;
    CWM_HIP_CHECK(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
    CWM_HIP_CHECK(hipStreamSynchronize(s));
    if (herr) {
        cwm_set_error("mask rows do not all have n_vis=%d visible tokens", n_vis);
        return CWM_ERR_MASK;
    }
    return CWM_OK;
}

extern "C" int cwm_unembed(const float* y_tokens_dev, const float* x_dev, const uint8_t* mask_dev, int B, int T, int C, int H, int W, int P,
                           int n_vis, float* out_dev, void* stream) {
    CWM_REQUIRE(y_tokens_dev && x_dev && mask_dev && out_dev, "cwm_unembed: null argument");
    CWM_REQUIRE(P > 0 && H % P == 0 && W % P == 0, "cwm_unembed: image size must be divisible by the patch size");
    hipStream_t s = (hipStream_t)stream;
    const int Nt = T * (H / P) * (W / P);
    Scratch sc;
    int* perm = sc.get<int>((size_t)B * Nt);
    int* rank = sc.get<int>((size_t)B * Nt);
    int* err = sc.get<int>(4, true);
    CWM_REQUIRE(perm && rank && err, "cwm_unembed: out of device memory");
    int rc;
    if ((rc = launch_mask_to_perm(mask_dev, B, Nt, n_vis, perm, err, s))) return rc;
    if ((rc = launch_perm_to_rank(perm, rank, B, Nt, s))) return rc;
    UnembedParams u;
    memset(&u, 0, sizeof(u));
    u.y = y_tokens_dev; u.x = x_dev; u.sb = (int64_t)T * C * H * W; u.st = (int64_t)C * H * W; u.sc = (int64_t)H * W;
    u.mask = mask_dev; u.rank = rank; u.B = B; u.T = T; u.C = C; u.H = H; u.W = W; u.P = P; u.n_vis = n_vis; u.Nm = Nt - n_vis; u.out = out_dev;
    if ((rc = launch_unembed(u, s))) return rc;
    int herr = 0;
    CWM_HIP_CHECK(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
    CWM_HIP_CHECK(hipStreamSynchronize(s));
    if (herr) {
        cwm_set_error("mask rows do not all have n_vis=%d visible tokens", n_vis);
        return CWM_ERR_MASK;
    }
    return CWM_OK;
}

extern "C" int cwm_mask_row_counts(const uint8_t* mask_dev, int B, int Nt, int32_t* counts_dev, void* stream) {
    CWM_REQUIRE(mask_dev && counts_dev && B > 0 && Nt > 0, "cwm_mask_row_counts: bad argument");
    return launch_mask_row_counts(mask_dev, B, Nt, counts_dev, (hipStream_t)stream);
}

extern "C" int cwm_mask_flip_picks(uint8_t* mask_dev, int B, int Nt, const int32_t* table_dev, int n_rows, void* stream) {
    CWM_REQUIRE(mask_dev && table_dev && B > 0 && Nt > 0 && n_rows >= 0 && n_rows <= B, "cwm_mask_flip_picks: bad argument");
    return launch_mask_flip_picks(mask_dev, Nt, table_dev, n_rows, (hipStream_t)stream);
}

extern "C" int cwm_prompt_table_expand(const int32_t* table_dev, int S, int T, int grid_h, int grid_w, int frame, uint8_t* active_dev, uint8_t* passive_dev,
                                       int32_t* shifts_dev, void* stream) {
    CWM_REQUIRE(table_dev && active_dev && passive_dev && shifts_dev, "cwm_prompt_table_expand: null argument");
    CWM_REQUIRE(S > 0 && T >= 2 && grid_h > 0 && grid_w > 0 && frame >= 1 && frame < T, "cwm_prompt_table_expand: bad sizes (S %d, T %d, grid %d x %d, frame %d)", S, T, grid_h, grid_w, frame);
    return launch_prompt_table_expand(table_dev, S, grid_h * grid_w, grid_w, T, frame, active_dev, passive_dev, shifts_dev, (hipStream_t)stream);
}

extern "C" int cwm_shift_prompts(const float* x_dev, int B, int T, int C, int H, int W, int P, int frame, int S, int fix_passive,
                                 const uint8_t* active_dev, const uint8_t* masks_dev, const int32_t* shifts_dev, float* x_out_dev,
                                 uint8_t* mask_out_dev, void* stream) {
    CWM_REQUIRE(active_dev && masks_dev && shifts_dev && (x_out_dev || mask_out_dev) && (x_dev || !x_out_dev), "cwm_shift_prompts: null argument");
    CWM_REQUIRE(B > 0 && S > 0 && T > 0 && C > 0 && P > 0, "cwm_shift_prompts: bad sizes");
    ShiftPromptParams p;
    memset(&p, 0, sizeof(p));
    p.x = x_dev; p.B = B; p.S = S; p.T = T; p.C = C; p.H = H; p.W = W; p.P = P; p.frame = frame; p.fix_passive = fix_passive;
    p.active = active_dev; p.masks = masks_dev; p.shifts = shifts_dev; p.x_out = x_out_dev; p.mask_out = mask_out_dev;
    return launch_shift_prompts(p, (hipStream_t)stream);
}

