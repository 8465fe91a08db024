// IMU-conditioned conjoined padded predictor behind the C ABI (BASELINE configs[4]).
// Restates `ConjoinedPaddedVisionTransformer.forward` for `imu400_base_4x4patch_2frames_1tube`
// (cwm/models/VideoMAE/conjoined_vmae.py:889-1011, 852-887, 1230-1243): two token streams (RGB "main",
// IMU "context"), null-token padding (:49-165), cross-attention blocks BEFORE encoder blocks 0,3,6,9 and
// AFTER every decoder block (:543-576, :688-720; cwm/models/transformer.py:253-378, 442-583).
#include <stddef.h>

#include "engine.h"

using namespace cwm;

namespace {

struct CrossW {
    float *n1_g, *n1_b, *n1s_g, *n1s_b, *n2_g, *n2_b, *n2s_g, *n2s_b;
    LinearW qk, qk_src, v, v_src, proj, proj_src, mlp_t0, mlp_t2, mlp_s0, mlp_s2;
};

struct StreamW {
    int enc_dim = 0, dec_dim = 0, enc_heads = 0, dec_heads = 0, n_tok = 0, max_pad = 0, out_dim = 0, embed_k = 0, embed_kpad = 0;
    std::vector<BlockW> enc, dec;
    LinearW embed, e2d, head;
    float *enc_norm_g = nullptr, *enc_norm_b = nullptr, *dec_norm_g = nullptr, *dec_norm_b = nullptr;
    float *mask_token = nullptr, *null_enc = nullptr;
    float *pos_enc_ext = nullptr, *pos_dec_ext = nullptr;  // [n_tok + max_pad][D]; pad rows: 0 (enc) / null_token_dec (dec)
    // workspace
    uint8_t* ext_mask = nullptr;
    int* perm = nullptr;
    bf16* tokens_in = nullptr;
    float *x_enc = nullptr, *x_dec = nullptr;
    StreamBuffers sb;
};

}  // namespace

struct cwm_conj_model {
    Engine eng;
    cwm_conj_config cfg;
    StreamW main, ctx;
    std::vector<CrossW> enc_cross, dec_cross;
    // workspace shared by the cross blocks
    int ws_batch = 0, ws_vmain = 0, ws_vctx = 0;
    int* err = nullptr;
    float *qk = nullptr, *v = nullptr, *qk_src = nullptr, *v_src = nullptr, *scores_t = nullptr, *cross_partial = nullptr;
    bf16 *ybuf = nullptr, *ysbuf = nullptr;
    // batch lanes (cwm_conj_set_lanes; see cwm_model in model.hip)
    int lanes = 2;
    hipStream_t lane_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // the context (IMU) stream of every lane runs its blocks on a stream of its own between two cross blocks (conj_forward_lane)
    hipStream_t ctx_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_ctx[2] = {nullptr, nullptr}, ev_main[2] = {nullptr, nullptr};
    hipEvent_t ev_cross[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // run_cross
    ~cwm_conj_model() {
        if (lane_stream) (void)hipStreamDestroy(lane_stream);
        for (int i = 0; i < 2; ++i) {
            if (ctx_stream[i]) (void)hipStreamDestroy(ctx_stream[i]);
            if (ev_ctx[i]) (void)hipEventDestroy(ev_ctx[i]);
            if (ev_main[i]) (void)hipEventDestroy(ev_main[i]);
            for (int k = 0; k < 4; ++k)
                if (ev_cross[i][k]) (void)hipEventDestroy(ev_cross[i][k]);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
};

namespace {

// One lane's view of the model: the weight pointers of both streams (copied), and the slice of every batch-major workspace buffer
// that lies behind the capacity of the b0 batch elements of the lanes before it.
struct ConjLane {
    cwm_conj_model* m;
    StreamW main, ctx;
    int* err;
    float *qk, *v, *qk_src, *v_src, *scores_t, *cross_partial;
    bf16 *ybuf, *ysbuf;
    bool have_prev = false;  // a cross block of this forward has used the projection buffers (run_cross); carried from stage to stage
};

void shift_stream(StreamW& S, int b0, int vcap, int mlp_ratio, bool small) {
    const int next = S.n_tok + S.max_pad;
    const size_t rows_e = (size_t)b0 * vcap, rows_d = (size_t)b0 * next;
    const size_t act = std::max(rows_e * S.enc_dim, rows_d * S.dec_dim);
    S.ext_mask += rows_d;
    S.perm += rows_d;
    S.tokens_in += 2 * rows_e * S.embed_kpad;
    S.x_enc += rows_e * S.enc_dim;
    S.x_dec += rows_d * S.dec_dim;
    S.sb.hbuf += 2 * act;
    S.sb.gbuf += 2 * act * mlp_ratio;
    if (small) {
        S.sb.qkv_f32 += 3 * act;
    } else {
        S.sb.qbuf += 2 * act;
        S.sb.kbuf += 2 * act;
        S.sb.vbuf += 2 * act;
    }
}

// scratch of the context-side partials: the larger of the two kernel forms' needs (conj_kernels.hip / conj_attention.hip)
size_t cross_partial_floats(int B, int heads, int M, int head_dim) {
    return std::max(cross_attention_partial_floats(B, heads, M, head_dim), cross_attention_mfma_partial_floats(B, heads, M, head_dim));
}

ConjLane conj_lane(cwm_conj_model* m, int lane, int b0) {
    ConjLane L;
    L.m = m;
    L.main = m->main;
    L.ctx = m->ctx;
    shift_stream(L.main, b0, m->ws_vmain, m->cfg.main.mlp_ratio, false);
    shift_stream(L.ctx, b0, m->ws_vctx, m->cfg.main.mlp_ratio, true);
    const size_t Nb = (size_t)b0 * (m->main.n_tok + m->main.max_pad), Mb = (size_t)b0 * (m->ctx.n_tok + m->ctx.max_pad);
    const size_t Dmax = std::max(m->main.enc_dim, m->main.dec_dim);
    const int Mtok = m->ctx.n_tok + m->ctx.max_pad;
    L.err = m->err + lane;
    L.qk = m->qk + Nb * 2 * Dmax;
    L.v = m->v + Nb * Dmax;
    L.qk_src = m->qk_src + Mb * 2 * Dmax;
    L.v_src = m->v_src + Mb * Dmax;
    L.scores_t = m->scores_t + Nb * m->cfg.cross_heads * Mtok;
    L.cross_partial = m->cross_partial + cross_partial_floats(b0, m->cfg.cross_heads, Mtok, (int)Dmax / m->cfg.cross_heads);
    L.ybuf = m->ybuf + 2 * Nb * Dmax;
    L.ysbuf = m->ysbuf + 2 * Mb * Dmax;
    return L;
}

}  // namespace

namespace {

int make_stream(Engine& E, StreamW& S, const std::string& pre, int embed_k, std::vector<int64_t> embed_shape, int depth_e, int depth_d,
                int mlp_ratio, bool sinusoid_f64) {
    int rc;
    S.embed_k = embed_k;
    S.embed_kpad = round_up(embed_k, 64);
    if ((rc = E.make_linear(S.embed, S.enc_dim, embed_k, true))) return rc;
    E.add_matrix_slot(pre + "encoder.patch_embed.proj.weight", &S.embed, embed_shape);
    E.add_vec_slot(pre + "encoder.patch_embed.proj.bias", S.embed.bias, {S.enc_dim});
    S.enc.resize(depth_e);
    S.dec.resize(depth_d);
    for (int i = 0; i < depth_e; ++i)
        if ((rc = E.make_block(S.enc[i], pre + "encoder.blocks." + std::to_string(i) + ".", S.enc_dim, mlp_ratio * S.enc_dim))) return rc;
    if ((rc = E.make_vec(&S.enc_norm_g, S.enc_dim)) || (rc = E.make_vec(&S.enc_norm_b, S.enc_dim))) return rc;
    E.add_vec_slot(pre + "encoder.norm.weight", S.enc_norm_g, {S.enc_dim});
    E.add_vec_slot(pre + "encoder.norm.bias", S.enc_norm_b, {S.enc_dim});
    if ((rc = E.make_linear(S.e2d, S.dec_dim, S.enc_dim, false))) return rc;
    E.add_matrix_slot(pre + "encoder_to_decoder.weight", &S.e2d, {S.dec_dim, S.enc_dim});
    for (int i = 0; i < depth_d; ++i)
        if ((rc = E.make_block(S.dec[i], pre + "decoder.blocks." + std::to_string(i) + ".", S.dec_dim, mlp_ratio * S.dec_dim))) return rc;
    if ((rc = E.make_vec(&S.dec_norm_g, S.dec_dim)) || (rc = E.make_vec(&S.dec_norm_b, S.dec_dim))) return rc;
    E.add_vec_slot(pre + "decoder.norm.weight", S.dec_norm_g, {S.dec_dim});
    E.add_vec_slot(pre + "decoder.norm.bias", S.dec_norm_b, {S.dec_dim});
    if ((rc = E.make_linear(S.head, S.out_dim, S.dec_dim, true))) return rc;
    E.add_matrix_slot(pre + "decoder.head.weight", &S.head, {S.out_dim, S.dec_dim});
    E.add_vec_slot(pre + "decoder.head.bias", S.head.bias, {S.out_dim});
    if ((rc = E.make_vec(&S.mask_token, S.dec_dim)) || (rc = E.make_vec(&S.null_enc, S.enc_dim))) return rc;
    E.add_vec_slot(pre + "mask_token", S.mask_token, {1, 1, S.dec_dim});
    E.add_vec_slot(pre + "null_token_enc", S.null_enc, {1, 1, S.enc_dim});
    // positional tables with max_pad extra rows; the decoder's pad rows hold null_token_dec (_pad_pos_embed :154-165)
    if (sinusoid_f64) {
        if ((rc = E.make_sinusoid(&S.pos_enc_ext, S.n_tok, S.enc_dim, S.max_pad)) || (rc = E.make_sinusoid(&S.pos_dec_ext, S.n_tok, S.dec_dim, S.max_pad)))
            return rc;
    } else {
        if ((rc = E.make_pos_embedding_f32(&S.pos_enc_ext, S.n_tok, S.enc_dim, S.max_pad)) ||
            (rc = E.make_pos_embedding_f32(&S.pos_dec_ext, S.n_tok, S.dec_dim, S.max_pad)))
            return rc;
    }
    E.add_vec_slot(pre + "null_token_dec", S.pos_dec_ext + (size_t)S.n_tok * S.dec_dim, {1, 1, S.dec_dim}, S.max_pad);
    return 0;
}

int make_cross(Engine& E, CrossW& C, const std::string& pre, int ci, int cs, int ratio) {
    int rc;
    float** vecs[8] = {&C.n1_g, &C.n1_b, &C.n1s_g, &C.n1s_b, &C.n2_g, &C.n2_b, &C.n2s_g, &C.n2s_b};
    const int dims[8] = {ci, ci, cs, cs, ci, ci, cs, cs};
    const char* names[8] = {"norm1_cross.weight", "norm1_cross.bias", "norm1_src_cross.weight", "norm1_src_cross.bias",
                            "norm2.weight", "norm2.bias", "norm2_src.weight", "norm2_src.bias"};
    for (int i = 0; i < 8; ++i) {
        if ((rc = E.make_vec(vecs[i], dims[i]))) return rc;
        E.add_vec_slot(pre + names[i], *vecs[i], {dims[i]});
    }
    const int D = ci;
    if ((rc = E.make_linear(C.qk, 2 * D, ci, false)) || (rc = E.make_linear(C.qk_src, 2 * D, cs, false)) || (rc = E.make_linear(C.v, D, ci, false)) ||
        (rc = E.make_linear(C.v_src, D, cs, false)) || (rc = E.make_linear(C.proj, ci, D, true)) || (rc = E.make_linear(C.proj_src, cs, D, true)) ||
        (rc = E.make_linear(C.mlp_t0, ratio * ci, ci, true)) || (rc = E.make_linear(C.mlp_t2, ci, ratio * ci, true)) ||
        (rc = E.make_linear(C.mlp_s0, ratio * cs, cs, true)) || (rc = E.make_linear(C.mlp_s2, cs, ratio * cs, true)))
        return rc;
    E.add_matrix_slot(pre + "cross_attention.qk.weight", &C.qk, {2 * D, ci});
    E.add_matrix_slot(pre + "cross_attention.qk_src.weight", &C.qk_src, {2 * D, cs});
    E.add_matrix_slot(pre + "cross_attention.v.weight", &C.v, {D, ci});
    E.add_matrix_slot(pre + "cross_attention.v_src.weight", &C.v_src, {D, cs});
    E.add_matrix_slot(pre + "cross_attention.projection.weight", &C.proj, {ci, D});
    E.add_vec_slot(pre + "cross_attention.projection.bias", C.proj.bias, {ci});
    E.add_matrix_slot(pre + "cross_attention.projection_src.weight", &C.proj_src, {cs, D});
    E.add_vec_slot(pre + "cross_attention.projection_src.bias", C.proj_src.bias, {cs});
    E.add_matrix_slot(pre + "mlp.trg.layers.0.weight", &C.mlp_t0, {ratio * ci, ci});
    E.add_vec_slot(pre + "mlp.trg.layers.0.bias", C.mlp_t0.bias, {ratio * ci});
    E.add_matrix_slot(pre + "mlp.trg.layers.2.weight", &C.mlp_t2, {ci, ratio * ci});
    E.add_vec_slot(pre + "mlp.trg.layers.2.bias", C.mlp_t2.bias, {ci});
    E.add_matrix_slot(pre + "mlp.src.layers.0.weight", &C.mlp_s0, {ratio * cs, cs});
    E.add_vec_slot(pre + "mlp.src.layers.0.bias", C.mlp_s0.bias, {ratio * cs});
    E.add_matrix_slot(pre + "mlp.src.layers.2.weight", &C.mlp_s2, {cs, ratio * cs});
    E.add_vec_slot(pre + "mlp.src.layers.2.bias", C.mlp_s2.bias, {cs});
    return 0;
}

int stream_workspace(Engine& E, StreamW& S, int B, int vmax, int mlp_ratio, bool small) {
    const int next = S.n_tok + S.max_pad;
    const size_t rows_e = (size_t)B * vmax, rows_d = (size_t)B * next;
    int rc;
    if ((rc = E.ws(&S.ext_mask, rows_d)) || (rc = E.ws(&S.perm, rows_d)) || (rc = E.ws(&S.tokens_in, 2 * rows_e * S.embed_kpad)) ||
        (rc = E.ws(&S.x_enc, rows_e * S.enc_dim)) || (rc = E.ws(&S.x_dec, rows_d * S.dec_dim)))
        return rc;
    const size_t act = std::max(rows_e * S.enc_dim, rows_d * S.dec_dim);
    if ((rc = E.ws(&S.sb.hbuf, 2 * act)) || (rc = E.ws(&S.sb.gbuf, 2 * act * mlp_ratio))) return rc;
    if (small) {
        if ((rc = E.ws(&S.sb.qkv_f32, 3 * act))) return rc;
    } else {
        if ((rc = E.ws(&S.sb.qbuf, 2 * act)) || (rc = E.ws(&S.sb.kbuf, 2 * act))) return rc;
        if ((rc = E.ws(&S.sb.vbuf, 2 * act))) return rc;
    }
    return 0;
}

int ensure_workspace(cwm_conj_model* m, int B, int vmain, int vctx) {
    if (m->ws_batch > 0 && B <= m->ws_batch && vmain <= m->ws_vmain && vctx <= m->ws_vctx) return 0;
    Engine& E = m->eng;
    if (int rc = E.free_workspace()) return rc;
    const int Bc = std::max(B, m->ws_batch), vm = std::max(vmain, m->ws_vmain), vc = std::max(vctx, m->ws_vctx);
    int rc;
    if ((rc = stream_workspace(E, m->main, Bc, vm, m->cfg.main.mlp_ratio, false)) || (rc = stream_workspace(E, m->ctx, Bc, vc, m->cfg.main.mlp_ratio, true)))
        return rc;
    const size_t Nmax = (size_t)Bc * (m->main.n_tok + m->main.max_pad), Mmax = (size_t)Bc * (m->ctx.n_tok + m->ctx.max_pad);
    const size_t Dmax = std::max(m->main.enc_dim, m->main.dec_dim);
    const int Mtok = m->ctx.n_tok + m->ctx.max_pad;
    if ((rc = E.ws(&m->err, 4)) || (rc = E.ws(&m->qk, Nmax * 2 * Dmax)) || (rc = E.ws(&m->v, Nmax * Dmax)) || (rc = E.ws(&m->qk_src, Mmax * 2 * Dmax)) ||
        (rc = E.ws(&m->v_src, Mmax * Dmax)) || (rc = E.ws(&m->scores_t, Nmax * m->cfg.cross_heads * Mtok)) ||
        (rc = E.ws(&m->cross_partial, cross_partial_floats(Bc, m->cfg.cross_heads, Mtok, (int)Dmax / m->cfg.cross_heads))) || (rc = E.ws(&m->ybuf, 2 * Nmax * Dmax)) ||
        (rc = E.ws(&m->ysbuf, 2 * Mmax * Dmax)))
        return rc;
    m->ws_batch = Bc;
    m->ws_vmain = vm;
    m->ws_vctx = vc;
    // (the zero fills above ran on the null stream; the lane streams are non-blocking and would not wait for them)
    CWM_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

int layernorm_to(Engine& E, const float* x, int rows, int D, const float* g, const float* b, bf16* out, int planes, hipStream_t s) {
    LayerNormParams ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = x; ln.ldx = D; ln.gamma = g; ln.beta = b; ln.eps = E.ln_eps; ln.D = D; ln.rows = rows;
    ln.out = out; ln.out_plane = (int64_t)rows * D; ln.ldo = D;
    return E.run_layernorm(ln, planes, s);
}

// y = A W^T in the GEMM A-operand layout (what the MFMA cross attention reads its token fragments from)
int linear_operand(Engine& E, const bf16* A, int rows, int K, const LinearW& L, bf16* out, int planes, hipStream_t s) {
    GemmParams g = gemm_base(A, K, L, rows, planes);
    g.epi = EPI_BF16; g.out_hi = out; g.out_plane = (int64_t)rows * L.N; g.ldo = L.N;
    return E.run_gemm(g, planes, s);
}

int linear_f32(Engine& E, const bf16* A, int rows, int K, const LinearW& L, float* C, const float* resid, int planes, hipStream_t s) {
    GemmParams g = gemm_base(A, K, L, rows, planes);
    g.epi = EPI_F32; g.C = C; g.ldc = L.N; g.resid = resid; g.ldr = L.N;
    return E.run_gemm(g, planes, s);
}

int linear_gelu(Engine& E, const bf16* A, int rows, int K, const LinearW& L, bf16* out, int planes, hipStream_t s) {
    GemmParams g = gemm_base(A, K, L, rows, planes);
    g.epi = EPI_BF16_GELU; g.out_hi = out; g.out_plane = (int64_t)rows * L.N; g.ldo = L.N;
    return E.run_gemm(g, planes, s);
}

// CrossAttentionTransformerBlock.forward (transformer.py:559-583) with with_self_attention=False.
//
// s = the lane's stream (RGB side), sc = the lane's context stream (== s: everything in order on one stream).  With two streams and the
// MFMA kernels the block runs as two chains that meet once:
//     s :  LN(x) -> qk, v projections ---+--> role A (y = softmax_M . v_src) -> x += proj(y) -> LN2 -> MLP_trg
//     sc:  LN(src) -> qk_src, v_src -----+--> role B + combine (y_src = softmax_N . v) -> src += proj_src(y_src) -> LN2 -> MLP_src
// (+: each stream waits for the other one's projections).  The two roles stream disjoint halves of the RGB-side projections and run
// concurrently; the ~10 tiny context-side launches leave the lane's stream.  ev[0..3] = {s projections done, sc projections done, role A
// done, role B done}: the last two guard the projection buffers against the NEXT cross block of the lane (role B reads qk / v that the
// next block's projections on s overwrite, role A reads qk_src / v_src that the next block overwrites on sc).
int run_cross(ConjLane& L, const CrossW& C, float* x, int N, int ci, float* src, int M, int cs, int B, int planes, hipStream_t s, hipStream_t sc,
              hipEvent_t* ev, bool& have_prev) {
    cwm_conj_model* m = L.m;
    Engine& E = m->eng;
    const int D = ci, heads = m->cfg.cross_heads, hd = D / heads;
    const int rows = B * N, rows_s = B * M;
    int rc;
    // main-stream projections: operand layout (bf16 hi [, lo] planes, the same 4 bytes per element as fp32) for the MFMA kernel
    const bool mfma = E.tune.conj_attn && cross_attention_mfma_ok(hd, M) && cross_attention_mfma_fits(B, N, heads, hd) && (2 * D) % 32 == 0;
    const bool two = sc != s && mfma;
    if (sc != s && !two) {  // VALU fallback: one chain on s, bracketed by the context stream
        CWM_HIP_CHECK(hipEventRecord(ev[1], sc));
        CWM_HIP_CHECK(hipStreamWaitEvent(s, ev[1], 0));
    }
    hipStream_t const t = two ? sc : s;  // where the context side runs
    if (two && have_prev) {
        CWM_HIP_CHECK(hipStreamWaitEvent(s, ev[3], 0));  // the previous block's role B has read qk / v
        CWM_HIP_CHECK(hipStreamWaitEvent(sc, ev[2], 0));  // ... its role A has read qk_src / v_src
    }
    if ((rc = layernorm_to(E, x, rows, ci, C.n1_g, C.n1_b, L.main.sb.hbuf, planes, s))) return rc;
    if (mfma) {
        if ((rc = linear_operand(E, L.main.sb.hbuf, rows, ci, C.qk, reinterpret_cast<bf16*>(L.qk), planes, s))) return rc;
        if ((rc = linear_operand(E, L.main.sb.hbuf, rows, ci, C.v, reinterpret_cast<bf16*>(L.v), planes, s))) return rc;
    } else {
        if ((rc = linear_f32(E, L.main.sb.hbuf, rows, ci, C.qk, L.qk, nullptr, planes, s))) return rc;
        if ((rc = linear_f32(E, L.main.sb.hbuf, rows, ci, C.v, L.v, nullptr, planes, s))) return rc;
    }
    if ((rc = layernorm_to(E, src, rows_s, cs, C.n1s_g, C.n1s_b, L.ctx.sb.hbuf, planes, t))) return rc;
    if ((rc = linear_f32(E, L.ctx.sb.hbuf, rows_s, cs, C.qk_src, L.qk_src, nullptr, planes, t))) return rc;
    if ((rc = linear_f32(E, L.ctx.sb.hbuf, rows_s, cs, C.v_src, L.v_src, nullptr, planes, t))) return rc;
    if (two) {
        CWM_HIP_CHECK(hipEventRecord(ev[0], s));
        CWM_HIP_CHECK(hipEventRecord(ev[1], sc));
        CWM_HIP_CHECK(hipStreamWaitEvent(sc, ev[0], 0));
        CWM_HIP_CHECK(hipStreamWaitEvent(s, ev[1], 0));
    }
    CrossAttnParams ca;
    memset(&ca, 0, sizeof(ca));
    ca.qk = L.qk; ca.v = L.v; ca.qk_op = reinterpret_cast<const bf16*>(L.qk); ca.v_op = reinterpret_cast<const bf16*>(L.v); ca.qk_src = L.qk_src; ca.v_src = L.v_src; ca.B = B; ca.N = N; ca.M = M; ca.heads = heads; ca.head_dim = hd;
    ca.scale = 1.0f / sqrtf((float)hd);
    ca.y = L.ybuf; ca.y_plane = (int64_t)rows * D; ca.y_src = L.ysbuf; ca.y_src_plane = (int64_t)rows_s * D; ca.scores_t = L.scores_t; ca.partial = L.cross_partial;
    if (two) {
        if ((rc = launch_cross_attention_mfma_roles(ca, planes, s, sc, 1))) return rc;
        CWM_HIP_CHECK(hipEventRecord(ev[2], s));
        if ((rc = launch_cross_attention_mfma_roles(ca, planes, s, sc, 2))) return rc;
        CWM_HIP_CHECK(hipEventRecord(ev[3], sc));
        have_prev = true;
    } else {
        // both directions: 2 x (scores 2 N M hd + P.V 2 N M hd) FLOP per head
        if ((rc = E.timed(CWM_KCLASS_CROSS_ATTN, 8.0 * (double)B * heads * N * M * hd, s,
                          [&] { return mfma ? launch_cross_attention_mfma(ca, planes, s) : launch_cross_attention(ca, planes, s); })))
            return rc;
    }
    if ((rc = linear_f32(E, L.ybuf, rows, D, C.proj, x, x, planes, s))) return rc;          // x += proj(y) + b
    if ((rc = layernorm_to(E, x, rows, ci, C.n2_g, C.n2_b, L.main.sb.hbuf, planes, s))) return rc;
    if ((rc = linear_gelu(E, L.main.sb.hbuf, rows, ci, C.mlp_t0, L.main.sb.gbuf, planes, s))) return rc;
    if ((rc = linear_f32(E, L.main.sb.gbuf, rows, C.mlp_t0.N, C.mlp_t2, x, x, planes, s))) return rc;
    if ((rc = linear_f32(E, L.ysbuf, rows_s, D, C.proj_src, src, src, planes, t))) return rc;
    if ((rc = layernorm_to(E, src, rows_s, cs, C.n2s_g, C.n2s_b, L.ctx.sb.hbuf, planes, t))) return rc;
    if ((rc = linear_gelu(E, L.ctx.sb.hbuf, rows_s, cs, C.mlp_s0, L.ctx.sb.gbuf, planes, t))) return rc;
    if ((rc = linear_f32(E, L.ctx.sb.gbuf, rows_s, C.mlp_s0.N, C.mlp_s2, src, src, planes, t))) return rc;
    if (sc != s && !two) {  // VALU fallback: the context stream continues behind the whole block
        CWM_HIP_CHECK(hipEventRecord(ev[0], s));
        CWM_HIP_CHECK(hipStreamWaitEvent(sc, ev[0], 0));
    }
    return 0;
}

// tokens + pos | null tokens, gathered by the padded mask (pad_and_mask_input, conjoined_vmae.py:125-134)
int embed_stream(cwm_conj_model* m, StreamW& S, int B, int vmax, int planes, hipStream_t s) {
    Engine& E = m->eng;
    const int next = S.n_tok + S.max_pad;
    GemmParams g = gemm_base(S.tokens_in, S.embed_kpad, S.embed, B * vmax, planes);
    g.epi = EPI_F32; g.C = S.x_enc; g.ldc = S.enc_dim;
    g.resid = S.pos_enc_ext; g.ldr = S.enc_dim; g.resid_rowmap = S.perm; g.rows_in = vmax; g.rows_out = vmax; g.map_stride = next;
    if (int rc = E.run_gemm(g, planes, s)) return rc;
    return launch_fix_pad_rows(S.x_enc, S.perm, B, next, vmax, S.n_tok, S.enc_dim, S.null_enc, s);
}

// encoder.norm, encoder_to_decoder, [x_vis + pos_ext[vis] | mask_token + pos_ext[masked]]
int to_decoder(cwm_conj_model* m, StreamW& S, int B, int vmax, int planes, hipStream_t s) {
    Engine& E = m->eng;
    const int next = S.n_tok + S.max_pad;
    int rc;
    if ((rc = layernorm_to(E, S.x_enc, B * vmax, S.enc_dim, S.enc_norm_g, S.enc_norm_b, S.sb.hbuf, planes, s))) return rc;
    GemmParams g = gemm_base(S.sb.hbuf, S.enc_dim, S.e2d, B * vmax, planes);
    g.epi = EPI_F32; g.C = S.x_dec; g.ldc = S.dec_dim;
    g.resid = S.pos_dec_ext; g.ldr = S.dec_dim; g.resid_rowmap = S.perm; g.rows_in = vmax; g.rows_out = next; g.map_stride = next;
    if ((rc = E.run_gemm(g, planes, s))) return rc;
    return E.run_fill_mask_tokens(S.x_dec, S.mask_token, S.pos_dec_ext, S.perm, B, next, vmax, S.dec_dim, s);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int cwm_conj_create(const cwm_conj_config* cfg, cwm_conj_model** out) {
    CWM_REQUIRE(cfg && out, "cwm_conj_create: null argument");
    const cwm_conj_config& c = *cfg;
    const cwm_config& mc = c.main;
    CWM_REQUIRE(mc.patch > 0 && mc.patch % 4 == 0 && mc.img_h % mc.patch == 0 && mc.img_w % mc.patch == 0 && mc.img_w % 4 == 0, "bad image/patch size");
    CWM_REQUIRE(mc.enc_dim == 64 * mc.enc_heads && mc.dec_dim == 64 * mc.dec_heads, "main stream needs head_dim 64");
    CWM_REQUIRE(mc.enc_dim % 128 == 0 && mc.dec_dim % 128 == 0 && mc.enc_dim <= 1024, "main stream widths must be multiples of 128");
    CWM_REQUIRE(c.ctx_seq_len % c.ctx_tubelet == 0 && c.ctx_seq_len / c.ctx_tubelet + c.ctx_max_pad <= 64, "context stream: at most 64 tokens incl. padding");
    CWM_REQUIRE(c.ctx_enc_dim % c.ctx_enc_heads == 0 && c.ctx_dec_dim % c.ctx_dec_heads == 0 && c.ctx_enc_dim / c.ctx_enc_heads <= 64 &&
                    c.ctx_dec_dim / c.ctx_dec_heads <= 64, "context stream head_dim must be <= 64");
    CWM_REQUIRE(c.ctx_enc_dim % 16 == 0 && c.ctx_dec_dim % 16 == 0 && c.ctx_enc_dim <= 1024, "context widths must be multiples of 16");
    CWM_REQUIRE(c.cross_heads > 0 && mc.enc_dim / c.cross_heads <= 192 && mc.enc_dim % c.cross_heads == 0 && mc.dec_dim % c.cross_heads == 0, "cross attention head_dim must be <= 192");
    CWM_REQUIRE(c.n_enc_cross >= 0 && c.n_enc_cross <= 16 && c.n_dec_cross >= 0 && c.n_dec_cross <= 16, "too many conjoining blocks");
    cwm_conj_model* m = new cwm_conj_model();
    m->cfg = c;
    Engine& E = m->eng;
    E.ln_eps = mc.ln_eps;
    CWM_HIP_CHECK(hipGetDevice(&E.device));
    StreamW& A = m->main;
    A.enc_dim = mc.enc_dim; A.dec_dim = mc.dec_dim; A.enc_heads = mc.enc_heads; A.dec_heads = mc.dec_heads;
    A.n_tok = (mc.img_h / mc.patch) * (mc.img_w / mc.patch) * mc.num_frames; A.max_pad = c.main_max_pad; A.out_dim = mc.in_chans * mc.patch * mc.patch;
    StreamW& S = m->ctx;
    S.enc_dim = c.ctx_enc_dim; S.dec_dim = c.ctx_dec_dim; S.enc_heads = c.ctx_enc_heads; S.dec_heads = c.ctx_dec_heads;
    S.n_tok = c.ctx_seq_len / c.ctx_tubelet; S.max_pad = c.ctx_max_pad; S.out_dim = c.ctx_in_chans * c.ctx_tubelet;
    int rc = 0;
    do {
        if ((rc = make_stream(E, A, "main_stream.", mc.in_chans * mc.patch * mc.patch, {mc.enc_dim, mc.in_chans, 1, mc.patch, mc.patch}, mc.enc_depth,
                              mc.dec_depth, mc.mlp_ratio, true)))
            break;
        if ((rc = make_stream(E, S, "context_stream.", c.ctx_in_chans * c.ctx_tubelet, {c.ctx_enc_dim, c.ctx_in_chans, c.ctx_tubelet, 1, 1}, mc.enc_depth,
                              mc.dec_depth, mc.mlp_ratio, false)))
            break;
        // loaded from the checkpoints but never used on this path (vmae.py:368-369)
        E.add_ignored_slot("context_stream.pos_embed_encoder.weight", {c.ctx_dec_dim, 2 * c.ctx_dec_dim});
        E.add_ignored_slot("context_stream.pos_embed_encoder.bias", {c.ctx_dec_dim});
        m->enc_cross.resize(c.n_enc_cross);
        m->dec_cross.resize(c.n_dec_cross);
        for (int i = 0; i < c.n_enc_cross && !rc; ++i) {
            const std::string k = std::to_string(c.enc_cross[i]);
            rc = make_cross(E, m->enc_cross[i], "encoder_conjoining_blocks." + k + "-" + k + ".", mc.enc_dim, c.ctx_enc_dim, c.cross_mlp_ratio);
        }
        for (int i = 0; i < c.n_dec_cross && !rc; ++i) {
            const std::string k = std::to_string(c.dec_cross[i]);
            rc = make_cross(E, m->dec_cross[i], "decoder_conjoining_blocks." + k + "-" + k + ".", mc.dec_dim, c.ctx_dec_dim, c.cross_mlp_ratio);
        }
    } while (0);
    if (rc) {
        delete m;
        return rc;
    }
    *out = m;
    return CWM_OK;
}

extern "C" void cwm_conj_destroy(cwm_conj_model* m) { delete m; }

extern "C" int cwm_conj_load_weight(cwm_conj_model* m, const char* key, const float* data, int on_device, const int64_t* shape, int ndim) {
    CWM_REQUIRE(m, "cwm_conj_load_weight: null model");
    return m->eng.load_weight(key, data, on_device, shape, ndim);
}

extern "C" int cwm_conj_missing_weights(cwm_conj_model* m, char* buf, int buflen) { return m->eng.missing_weights(buf, buflen); }

// One lane: batch elements [b0, b0 + B) of the call on stream s.
//
// The context (IMU) stream is a chain of ~300 tiny launches (25 / 50 tokens per sample: 5-15 us each, latency-bound) that only meets
// the RGB stream in the 8 cross blocks.  It runs on a stream of its own (sc), so that the chain hides under the RGB stream's kernels
// instead of extending the lane by ~4 %; inside a cross block the two streams exchange their projections once (run_cross).
// (Not while kernel timers are on: those want every launch alone on the chip.)
// Stages [stage_lo, stage_hi) of the lane's launch sequence: 0 = masks + tokenisation + embedding of both streams; 1 .. Le = encoder step i (the cross block in front of
// block i, then block i of both streams); Le + 1 = the two to_decoder steps; Le + 2 .. Le + 1 + Ld = decoder step i; Le + Ld + 2 = outputs.  cwm_conj_forward issues stage by
// stage over the lanes, as cwm_forward does: a forward is ~1000 launches per lane, and a host that issues lane 0 to its end first starts lane 1 that much later -- 1.5 ms of
// 61 normally, but 22 ms under rocprofv3 (whose launches cost ~20 us each: the lanes of the round-4 trace overlapped for half of their time only).  Measured without the
// profiler: no difference (61.2 ms per step either way; this model's step is the sum of its kernels -- 62.6 ms of kernel time in the one-lane trace).
static int conj_forward_lane(ConjLane& L, const cwm_conj_forward_args* a, int b0, int B, hipStream_t s, int lane, int stage_lo, int stage_hi) {
    cwm_conj_model* m = L.m;
    const cwm_conj_config& c = m->cfg;
    const cwm_config& mc = c.main;
    StreamW& A = L.main;
    StreamW& S = L.ctx;
    const int vm = a->n_vis_max, vc = a->n_vis_ctx_max;
    const int Nx = A.n_tok + A.max_pad, Mx = S.n_tok + S.max_pad;
    const int n_out = Nx - vm;
    Engine& E = m->eng;
    const int planes = a->mode == CWM_MODE_PARITY ? 2 : 1;
    const float* x_in = a->x_dev + (int64_t)b0 * a->x_stride_b;
    const uint8_t* mask_in = a->mask_dev + (size_t)b0 * A.n_tok;
    const float* ctx_in = a->ctx_dev + (size_t)b0 * c.ctx_in_chans * c.ctx_seq_len;
    const uint8_t* ctx_mask_in = a->ctx_mask_dev + (size_t)b0 * S.n_tok;
    float* y_tokens = a->y_tokens_dev + (size_t)b0 * n_out * A.out_dim;
    int rc;
    bool side = E.tune.conj_ctx_stream != 0;  // (0 keeps the context stream's blocks on the lane's own stream)
    for (int k = 0; k < CWM_KCLASS_COUNT; ++k) side = side && !E.timers[k].enabled;
    if (side && !m->ctx_stream[lane]) {
        CWM_HIP_CHECK(hipStreamCreateWithFlags(&m->ctx_stream[lane], hipStreamNonBlocking));
        CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_ctx[lane], hipEventDisableTiming));
        CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_main[lane], hipEventDisableTiming));
        for (int k = 0; k < 4; ++k) CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_cross[lane][k], hipEventDisableTiming));
    }
    bool& have_prev = L.have_prev;
    auto in_range = [&](int st) { return st >= stage_lo && st < stage_hi; };
    const int st_todec = mc.enc_depth + 1, st_out = mc.enc_depth + mc.dec_depth + 2;
    hipStream_t sc = side ? m->ctx_stream[lane] : s;
    // main -> ctx: the context stream may continue once everything queued on s so far is done; ctx -> main likewise
    auto ctx_follows_main = [&]() -> int {
        if (!side) return 0;
        CWM_HIP_CHECK(hipEventRecord(m->ev_main[lane], s));
        CWM_HIP_CHECK(hipStreamWaitEvent(sc, m->ev_main[lane], 0));
        return 0;
    };
    auto main_follows_ctx = [&]() -> int {
        if (!side) return 0;
        CWM_HIP_CHECK(hipEventRecord(m->ev_ctx[lane], sc));
        CWM_HIP_CHECK(hipStreamWaitEvent(s, m->ev_ctx[lane], 0));
        return 0;
    };

    if (in_range(0)) {
    have_prev = false;
    // a13: padded masks -> permutations [visible slots ascending | masked slots ascending] over n_tok + max_pad slots
    CWM_HIP_CHECK(hipMemsetAsync(L.err, 0, sizeof(int), s));
    if ((rc = launch_pad_mask(mask_in, B, A.n_tok, A.max_pad, vm, A.ext_mask, s))) return rc;
    if ((rc = launch_mask_to_perm(A.ext_mask, B, Nx, vm, A.perm, L.err, s))) return rc;
    if ((rc = launch_pad_mask(ctx_mask_in, B, S.n_tok, S.max_pad, vc, S.ext_mask, s))) return rc;
    if ((rc = launch_mask_to_perm(S.ext_mask, B, Mx, vc, S.perm, L.err, s))) return rc;

    // a2/a14: tokenise both streams (visible slots only)
    PatchGatherParams pg;
    memset(&pg, 0, sizeof(pg));
    pg.x = x_in; pg.sb = a->x_stride_b; pg.sc = a->x_stride_c; pg.st = a->x_stride_t; pg.normalize = a->normalize;
    pg.C = mc.in_chans; pg.H = mc.img_h; pg.W = mc.img_w; pg.P = mc.patch; pg.perm = A.perm; pg.Nt = A.n_tok; pg.perm_stride = Nx; pg.n_rows = vm; pg.B = B;
    pg.out = A.tokens_in; pg.out_plane = (int64_t)B * vm * A.embed_kpad; pg.ld = A.embed_kpad;
    if ((rc = E.run_patch_gather(pg, planes, s))) return rc;
    if ((rc = embed_stream(m, A, B, vm, planes, s))) return rc;
    ImuGatherParams ig;
    memset(&ig, 0, sizeof(ig));
    ig.imu = ctx_in; ig.B = B; ig.C = c.ctx_in_chans; ig.L = c.ctx_seq_len; ig.tubelet = c.ctx_tubelet; ig.perm = S.perm; ig.perm_stride = Mx;
    ig.n_rows = vc; ig.n_real = S.n_tok; ig.out = S.tokens_in; ig.out_plane = (int64_t)B * vc * S.embed_kpad; ig.ld = S.embed_kpad;
    if ((rc = launch_imu_gather(ig, planes, s))) return rc;
    if ((rc = embed_stream(m, S, B, vc, planes, s))) return rc;
    if ((rc = ctx_follows_main())) return rc;
    }

    // encoder: cross block BEFORE the self-attention blocks listed in enc_cross (forward_encoder_blocks :543-576)
    for (int i = 0; i < mc.enc_depth; ++i) {
        if (!in_range(1 + i)) continue;
        for (int k = 0; k < c.n_enc_cross; ++k)
            if (c.enc_cross[k] == i &&
                (rc = run_cross(L, m->enc_cross[k], A.x_enc, vm, A.enc_dim, S.x_enc, vc, S.enc_dim, B, planes, s, sc, m->ev_cross[lane], have_prev)))
                return rc;
        if ((rc = E.run_block(A.enc[i], A.x_enc, B, vm, A.enc_dim, A.enc_heads, planes, A.sb, s))) return rc;
        if ((rc = E.run_block_small(S.enc[i], S.x_enc, B, vc, S.enc_dim, S.enc_heads, planes, S.sb, sc))) return rc;
    }
    if (in_range(st_todec) && ((rc = to_decoder(m, A, B, vm, planes, s)) || (rc = to_decoder(m, S, B, vc, planes, sc)))) return rc;

    // decoder: cross block AFTER the blocks listed in dec_cross (forward_decoder_blocks :688-720)
    for (int i = 0; i < mc.dec_depth; ++i) {
        if (!in_range(st_todec + 1 + i)) continue;
        if ((rc = E.run_block(A.dec[i], A.x_dec, B, Nx, A.dec_dim, A.dec_heads, planes, A.sb, s))) return rc;
        if ((rc = E.run_block_small(S.dec[i], S.x_dec, B, Mx, S.dec_dim, S.dec_heads, planes, S.sb, sc))) return rc;
        for (int k = 0; k < c.n_dec_cross; ++k)
            if (c.dec_cross[k] == i &&
                (rc = run_cross(L, m->dec_cross[k], A.x_dec, Nx, A.dec_dim, S.x_dec, Mx, S.dec_dim, B, planes, s, sc, m->ev_cross[lane], have_prev)))
                return rc;
    }
    if (!in_range(st_out)) return CWM_OK;
    // context output (forward(..., output_context=True), conjoined_decode :990-1002): head_ctx(norm_ctx(x_c[:, -n_out_c:])) * ~null_mask_ctx
    if (a->y_ctx_tokens_dev) {
        const int n_out_c = Mx - vc;
        float* y_ctx = a->y_ctx_tokens_dev + (size_t)b0 * n_out_c * S.out_dim;
        LayerNormParams lc;
        memset(&lc, 0, sizeof(lc));
        lc.x = S.x_dec; lc.ldx = S.dec_dim; lc.gamma = S.dec_norm_g; lc.beta = S.dec_norm_b; lc.eps = E.ln_eps; lc.D = S.dec_dim;
        lc.rows = B * n_out_c; lc.rows_out_per_b = n_out_c; lc.rows_in_per_b = Mx; lc.in_offset = vc;
        lc.out = S.sb.hbuf; lc.out_plane = (int64_t)B * n_out_c * S.dec_dim; lc.ldo = S.dec_dim;
        if ((rc = E.run_layernorm(lc, planes, sc))) return rc;
        GemmParams gc = gemm_base(S.sb.hbuf, S.dec_dim, S.head, B * n_out_c, planes);
        gc.epi = EPI_F32; gc.C = y_ctx; gc.ldc = S.out_dim;
        if ((rc = E.run_gemm(gc, planes, sc))) return rc;
        if ((rc = launch_zero_pad_out_rows(y_ctx, S.perm, B, Mx, vc, n_out_c, S.n_tok, S.out_dim, sc))) return rc;
    }
    if ((rc = main_follows_ctx())) return rc;  // the call's stream semantics cover the context stream's work too

    // main output: head(norm(x[:, -n_out:])) * ~null_mask   (conjoined_decode :984-1002)
    LayerNormParams ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = A.x_dec; ln.ldx = A.dec_dim; ln.gamma = A.dec_norm_g; ln.beta = A.dec_norm_b; ln.eps = E.ln_eps; ln.D = A.dec_dim;
    ln.rows = B * n_out; ln.rows_out_per_b = n_out; ln.rows_in_per_b = Nx; ln.in_offset = vm;
    ln.out = A.sb.hbuf; ln.out_plane = (int64_t)B * n_out * A.dec_dim; ln.ldo = A.dec_dim;
    if ((rc = E.run_layernorm(ln, planes, s))) return rc;
    GemmParams g = gemm_base(A.sb.hbuf, A.dec_dim, A.head, B * n_out, planes);
    g.epi = EPI_F32; g.C = y_tokens; g.ldc = A.out_dim;
    if ((rc = E.run_gemm(g, planes, s))) return rc;
    return launch_zero_pad_out_rows(y_tokens, A.perm, B, Nx, vm, n_out, A.n_tok, A.out_dim, s);
}

extern "C" int cwm_conj_forward(cwm_conj_model* m, const cwm_conj_forward_args* a_in) {
    CWM_REQUIRE(m && a_in, "cwm_conj_forward: null argument");
    // the caller's struct may end before the fields later versions appended: copy what it has, the rest stays zero (= not requested)
    CWM_REQUIRE(a_in->struct_size >= offsetof(cwm_conj_forward_args, stream) + sizeof(void*) && a_in->struct_size <= 4096,
                "cwm_conj_forward: args->struct_size = %u is not a cwm_conj_forward_args (set it to sizeof(cwm_conj_forward_args))", a_in->struct_size);
    cwm_conj_forward_args a_copy;
    memset(&a_copy, 0, sizeof(a_copy));
    memcpy(&a_copy, a_in, std::min<size_t>(a_in->struct_size, sizeof(a_copy)));
    const cwm_conj_forward_args* a = &a_copy;
    if (int rc = cwm_require_device(m->eng.device, "cwm_conj_forward")) return rc;
    CWM_REQUIRE(a->x_dev && a->mask_dev && a->ctx_dev && a->ctx_mask_dev && a->y_tokens_dev, "cwm_conj_forward: x, mask, context, context mask and y are required");
    CWM_REQUIRE(a->mode == CWM_MODE_FAST || a->mode == CWM_MODE_PARITY, "cwm_conj_forward: bad mode %d", a->mode);
    const int B = a->batch, vm = a->n_vis_max, vc = a->n_vis_ctx_max;
    const int Nx = m->main.n_tok + m->main.max_pad;
    CWM_REQUIRE(B > 0 && vm > 0 && vm < Nx && vc > 0 && vc <= m->ctx.n_tok, "cwm_conj_forward: bad batch / visible counts (%d, %d, %d)", B, vm, vc);
    {
        char miss[256];
        const int nmiss = m->eng.missing_weights(miss, sizeof(miss));
        CWM_REQUIRE(nmiss == 0, "cwm_conj_forward: %d state-dict tensors not loaded (first: %s)", nmiss, miss);
    }
    if (int rc = ensure_workspace(m, B, vm, vc)) return rc;
    hipStream_t s = (hipStream_t)a->stream;

    // two lanes as in cwm_forward (model.hip): the halves share n_vis_max / n_vis_ctx_max, so the padded layout of every row is unchanged
    const bool two = m->lanes >= 2 && B >= 2 && (int64_t)(B / 2) * vm >= (m->eng.tune.min_lane_rows > 0 ? m->eng.tune.min_lane_rows : kMinLaneRowsConj);
    const int B0 = two ? (B + 1) / 2 : B;
    if (two) {
        if (!m->lane_stream) {
            CWM_HIP_CHECK(hipStreamCreateWithFlags(&m->lane_stream, hipStreamNonBlocking));
            CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
            CWM_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
        }
        CWM_HIP_CHECK(hipEventRecord(m->ev_fork, s));
        CWM_HIP_CHECK(hipStreamWaitEvent(m->lane_stream, m->ev_fork, 0));
    }
    ConjLane lanes[2] = {conj_lane(m, 0, 0), conj_lane(m, 1, two ? B0 : 0)};
    m->eng.overlapped = two;
    int rc = 0;
    // launch order: stage by stage over the lanes (conj_forward_lane), so that both queues fill at the same pace
    const int n_stages = m->cfg.main.enc_depth + m->cfg.main.dec_depth + 3;
    for (int st = 0; st < n_stages && !rc; ++st) {
        rc = conj_forward_lane(lanes[0], a, 0, B0, s, 0, st, st + 1);
        if (two && !rc) rc = conj_forward_lane(lanes[1], a, B0, B - B0, m->lane_stream, 1, st, st + 1);
    }
    m->eng.overlapped = 0;
    if (two) {  // join even after a failed launch: the caller's stream must not run ahead of work already queued on the lane
        CWM_HIP_CHECK(hipEventRecord(m->ev_join, m->lane_stream));
        CWM_HIP_CHECK(hipStreamWaitEvent(s, m->ev_join, 0));
    }
    if (rc) return rc;

    if (a->check) {
        int herr[2] = {0, 0};
        CWM_HIP_CHECK(hipMemcpyAsync(herr, m->err, (two ? 2 : 1) * sizeof(int), hipMemcpyDeviceToHost, s));
        CWM_HIP_CHECK(hipStreamSynchronize(s));
        if (herr[0] || herr[1]) {
            cwm_set_error("n_vis_max / n_vis_ctx_max do not match the masks (a row has more visible tokens, or the padding budget is exceeded)");
            return CWM_ERR_MASK;
        }
    }
    return CWM_OK;
}

extern "C" int cwm_conj_set_option(cwm_conj_model* m, const char* key, int value) {
    CWM_REQUIRE(m && key, "cwm_conj_set_option: null argument");
    const int rc = tuning_set_production(m->eng.tune, key, value);
    CWM_REQUIRE(rc != -2, "cwm_conj_set_option: gemm_debug bits 1, 2 and 8 are timing-only ablations (wrong outputs): development library only (cwm_debug_set)");
    CWM_REQUIRE(rc == 0, "cwm_conj_set_option: unknown option %s", key);
    return CWM_OK;
}

extern "C" int cwm_conj_set_lanes(cwm_conj_model* m, int lanes) {
    CWM_REQUIRE(m && (lanes == 1 || lanes == 2), "cwm_conj_set_lanes: lanes must be 1 or 2");
    m->lanes = lanes;
    return CWM_OK;
}

extern "C" int cwm_conj_timing_enable(cwm_conj_model* m, int kclass, int enable) {
    CWM_REQUIRE(m, "cwm_conj_timing_enable: null model");
    return m->eng.timing_enable(kclass, enable);
}

extern "C" int cwm_conj_timing_collect(cwm_conj_model* m, int kclass, cwm_kernel_stats* out) {
    CWM_REQUIRE(m, "cwm_conj_timing_collect: null model");
    return m->eng.timing_collect(kclass, out);
}
