// Whole-sequence decode over the transformer decoder (reference decoder/transform_decoder.py):
// the loop of predict_image_caption_sequence (recurrent_example.py:477-562) with the decoder
// swapped, run incrementally - one pass of the layer stack per new target position, keys and
// values of earlier positions cached per row, the encoder-side keys / values of every layer
// computed once per image.  The vocab projection, beam step and back-tracking are the kernels
// of the recurrent path.
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "icap_common.h"
#include "topk_epilogue.h"

namespace icap {

int dense_f32(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
int dense_f32_throughput(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
int dense_topk_half_tiles(int N);
int dense_topk_record_floats(int top_k);
bool dense_topk_supported(const float*, const float*, int, int, int, int);
int dense_topk_f32(const float*, const float*, const float*, int, int, int, int, int, float*, float*,
                   const LogitMasks*, hipStream_t);
bool beam_records_supported(int W, int half_tiles, int top_k);
int beam_step_from_records_f32(const float*, const float*, int, int, const float*, const uint8_t*,
                               const int32_t*, int, int, double, float*, int32_t*, int32_t*, float*,
                               uint8_t*, int32_t*, int, int, int, hipStream_t, int compact_first = 0);
int beam_step_f32(const float*, const float*, const uint8_t*, const int32_t*, int, int, int,
                  double, int, float*, int32_t*, int32_t*, float*, uint8_t*, int32_t*, int, int,
                  int, void*, size_t, hipStream_t);
size_t beam_step_workspace_bytes(int B, int W);
int backtrack(const int32_t*, const int32_t*, const float*, const uint8_t*, int32_t*, int32_t*,
              float*, int, int, int, hipStream_t);
int embed_pos_f32(const float*, const int32_t*, const float*, float*, int, int, int, hipStream_t);
int residual_layernorm_f32(float*, const float*, const float*, const float*, float*, int, int, float,
                           hipStream_t);
int mha_self_step_f32(const float*, const float*, float*, int, int, int, int, int, hipStream_t);
int add_f32(float*, const float*, long long, long long, hipStream_t);
int mha_cross_step_f32(const float*, const float*, float*, int, int, int, int, int, int, int,
                       const float*, int, hipStream_t);
int mha_scores_f32(const float*, const float*, const float*, float*, int, int, int, int, int, int,
                   hipStream_t);
int mha_combine_f32(const float*, const float*, float*, int, int, int, int, int, hipStream_t);

namespace {

__global__ void tfill_i32_kernel(int32_t* p, int32_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// dst[k][c0 + c] = src[k][c]: packs two [K,Ns] kernels side by side into one [K,Nd] operand.
__global__ void copy_columns_kernel(const float* __restrict__ src, float* __restrict__ dst, int K,
                                    int Ns, int Nd, int c0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)K * Ns) return;
  const int k = (int)(i / Ns), c = (int)(i - (long long)k * Ns);
  dst[(size_t)k * Nd + c0 + c] = src[i];
}

// [length, hidden] = [sin(pos * inv_timescale) | cos(pos * inv_timescale)] in fp32, as
// embedding/position_embedding.py:86-101 computes it (timescales 1 .. 1e4).
__global__ void position_table_kernel(float* __restrict__ table, int length, int hidden) {
  const int half = hidden / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= length * half) return;
  const int pos = i / half, j = i - pos * half;
  const float inc = (float)9.210340371976184 / ((float)half - 1.0f);  // log(1e4 / 1)
  const float inv = expf((float)j * -inc);
  const float t = (float)pos * inv;
  table[(size_t)pos * hidden + j] = sinf(t);
  table[(size_t)pos * hidden + half + j] = cosf(t);
}

struct TCarver {
  size_t off = 0;
  template <typename T>
  size_t take(size_t n) {
    off = (off + 255) / 256 * 256;
    const size_t o = off;
    off += n * sizeof(T);
    return o;
  }
};

#define ICAP_TTRY(expr)             \
  do {                              \
    const int rc_ = (expr);         \
    if (rc_ != ICAP_OK) return rc_; \
  } while (0)

}  // namespace

int position_table_f32(float* table, int length, int hidden, hipStream_t s) {
  ICAP_REQUIRE(table, "position_table_f32: null pointer");
  ICAP_REQUIRE(length > 0 && hidden >= 4 && hidden % 2 == 0,
               "position_table_f32: hidden must be even and at least 4");
  const int n = length * (hidden / 2);
  hipLaunchKernelGGL(position_table_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, table, length,
                     hidden);
  ICAP_LAUNCH_CHECK("position_table_kernel");
  return ICAP_OK;
}

}  // namespace icap

struct icap_tengine {
  icap_tconfig cfg;
  icap_tweights w;
  std::vector<icap_tlayer> layers;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  float *features, *x, *y, *q, *att, *o, *f, *pos_table, *logits;
  std::vector<float*> kv_self, kv_cross;       // per layer: [Tcap][N][2HD], [B*L][2HD]
  std::vector<float*> w_kv_self, b_kv_self;    // per layer: [hidden,2HD], [2HD]
  std::vector<float*> w_kv_cross, b_kv_cross;  // per layer: [D,2HD], [2HD]
  float *lp, *hist_scores, *hist_lp, *rec, *eos_x;
  int32_t *len, *hist_ids, *hist_parents, *start_ids;
  uint8_t* fin;
  char* beam_ws;
  size_t beam_ws_bytes;
  int topk = 0;
  unsigned timing = 0;
  struct Span { int cls; hipEvent_t a, b; };
  std::vector<Span> spans;
  size_t spans_used = 0;
};

namespace icap {
namespace {

// Brackets one launch with a pair of events from the pool that icap_tengine_set_timing
// reserved; a decode never creates events.  Launches that find the pool used up go unbracketed.
struct TTimed {
  icap_tengine* e;
  hipStream_t s;
  bool on;
  size_t idx = 0;
  TTimed(icap_tengine* e_, int cls, hipStream_t s_) : e(e_), s(s_), on(((e_->timing >> cls) & 1u) != 0) {
    if (!on) return;
    if (e->spans_used == e->spans.size()) {
      on = false;
      return;
    }
    idx = e->spans_used++;
    e->spans[idx].cls = cls;
    (void)hipEventRecord(e->spans[idx].a, s);
  }
  ~TTimed() {
    if (on) (void)hipEventRecord(e->spans[idx].b, s);
  }
};

int check_tconfig(const icap_tconfig& c) {
  ICAP_REQUIRE(c.B > 0 && c.W > 0 && c.L > 0 && c.F > 0 && c.D > 0 && c.hidden > 0 && c.heads > 0 &&
                   c.key_dim > 0 && c.filter > 0 && c.layers > 0 && c.V > 0 && c.T > 0,
               "tengine: every dimension must be positive");
  ICAP_REQUIRE(c.W <= ICAP_MAX_BEAM_WIDTH && c.W <= c.V, "tengine: beam width %d out of range", c.W);
  ICAP_REQUIRE(c.start_id >= 0 && c.start_id < c.V, "tengine: start_id outside the vocabulary");
  ICAP_REQUIRE(c.tile_mode == ICAP_TILE_REFERENCE || c.tile_mode == ICAP_TILE_REPEAT,
               "tengine: unknown tile_mode %d", c.tile_mode);
  ICAP_REQUIRE(c.mask_mode == ICAP_TMASK_AS_WRITTEN || c.mask_mode == ICAP_TMASK_CAUSAL,
               "tengine: unknown mask_mode %d", c.mask_mode);
  ICAP_REQUIRE(c.hidden % 4 == 0 && c.hidden <= 4096 && c.D % 4 == 0 && c.filter % 4 == 0,
               "tengine: hidden (<= 4096), D and filter must be multiples of 4");
  ICAP_REQUIRE(c.key_dim % 4 == 0, "tengine: key_dim must be a multiple of 4");
  if (c.precision != ICAP_PRECISION_FP32)
    return fail(ICAP_ERR_UNSUPPORTED, "tengine: only fp32 is built for the transformer path");
  ICAP_REQUIRE((long long)c.B * c.W * (long long)c.V < (1LL << 40), "tengine: logits too large");
  return ICAP_OK;
}

// One pass of the layer stack for target position `pos` (token ids of the rows in `ids`, null:
// token 0).  Leaves LayerNorm_out(x) of the new position in e->y.
int stack_pass(icap_tengine* e, const int32_t* ids, int pos, hipStream_t s) {
  const icap_tconfig& c = e->cfg;
  const int N = c.B * c.W, H = c.hidden, HD = c.heads * c.key_dim;
  const int stride_g = c.tile_mode == ICAP_TILE_REPEAT ? c.W : 1;
  const int stride_k = c.tile_mode == ICAP_TILE_REPEAT ? 1 : c.B;
  const float eps = 1e-6f;
  {
    TTimed t(e, ICAP_TK_EMBED, s);
    ICAP_TTRY(embed_pos_f32(e->w.embedding, ids, e->pos_table + (size_t)pos * H, e->x, N, c.V, H, s));
  }
  const float* pending = nullptr;  // sub-layer output not yet added to x
  for (int l = 0; l < c.layers; ++l) {
    const icap_tlayer& L = e->layers[l];
    {
      TTimed t(e, ICAP_TK_LAYERNORM, s);
      ICAP_TTRY(residual_layernorm_f32(e->x, pending, L.self_ln_gamma, L.self_ln_beta, e->y, N, H, eps, s));
    }
    {  // self-attention: query from LayerNorm(x); keys / values from the raw x as written
      TTimed t(e, ICAP_TK_PROJ, s);
      ICAP_TTRY(dense_f32_throughput(e->y, L.self_q_kernel, L.self_q_bias, e->q, N, H, HD, ICAP_ACT_NONE, s));
      float* kv_new = e->kv_self[l] + (size_t)pos * N * 2 * HD;
      ICAP_TTRY(dense_f32_throughput(c.kv_from_normed ? e->y : e->x, e->w_kv_self[l], e->b_kv_self[l], kv_new, N,
                          H, 2 * HD, ICAP_ACT_NONE, s));
    }
    {
      TTimed t(e, ICAP_TK_SELF_ATT, s);
      ICAP_TTRY(mha_self_step_f32(e->q, e->kv_self[l], e->att, N, c.heads, c.key_dim, pos, c.mask_mode, s));
    }
    {
      TTimed t(e, ICAP_TK_PROJ, s);
      ICAP_TTRY(dense_f32_throughput(e->att, L.self_o_kernel, L.self_o_bias, e->o, N, HD, H, ICAP_ACT_NONE, s));
    }
    {
      TTimed t(e, ICAP_TK_LAYERNORM, s);
      ICAP_TTRY(residual_layernorm_f32(e->x, e->o, L.cross_ln_gamma, L.cross_ln_beta, e->y, N, H, eps, s));
    }
    {
      TTimed t(e, ICAP_TK_PROJ, s);
      ICAP_TTRY(dense_f32_throughput(e->y, L.cross_q_kernel, L.cross_q_bias, e->q, N, H, HD, ICAP_ACT_NONE, s));
    }
    {
      TTimed t(e, ICAP_TK_CROSS_ATT, s);
      ICAP_TTRY(mha_cross_step_f32(e->q, e->kv_cross[l], e->att, c.B, c.W, stride_g, stride_k, c.L,
                                   c.heads, c.key_dim, nullptr, 0, s));
    }
    {
      TTimed t(e, ICAP_TK_PROJ, s);
      ICAP_TTRY(dense_f32_throughput(e->att, L.cross_o_kernel, L.cross_o_bias, e->o, N, HD, H, ICAP_ACT_NONE, s));
    }
    {
      TTimed t(e, ICAP_TK_LAYERNORM, s);
      ICAP_TTRY(residual_layernorm_f32(e->x, e->o, L.ffn_ln_gamma, L.ffn_ln_beta, e->y, N, H, eps, s));
    }
    {
      TTimed t(e, ICAP_TK_FFN, s);
      ICAP_TTRY(dense_f32_throughput(e->y, L.ffn_filter_kernel, L.ffn_filter_bias, e->f, N, H, c.filter,
                          ICAP_ACT_RELU, s));
      ICAP_TTRY(dense_f32_throughput(e->f, L.ffn_output_kernel, L.ffn_output_bias, e->o, N, c.filter, H,
                          ICAP_ACT_NONE, s));
    }
    pending = e->o;
  }
  TTimed t(e, ICAP_TK_LAYERNORM, s);
  return residual_layernorm_f32(e->x, pending, e->w.out_ln_gamma, e->w.out_ln_beta, e->y, N, H, eps, s);
}

int tdecode_impl(icap_tengine* e, const float* img, int32_t* tokens, float* best_score, hipStream_t s) {
  const icap_tconfig& c = e->cfg;
  const int N = c.B * c.W, BL = c.B * c.L, HD = c.heads * c.key_dim;
  {
    TTimed t(e, ICAP_TK_ENCODE, s);
    ICAP_TTRY(dense_f32(img, e->w.enc_kernel, e->w.enc_bias, e->features, BL, c.F, c.D, ICAP_ACT_RELU, s));
    // encoder-side keys and values of every layer, once per image
    for (int l = 0; l < c.layers; ++l)
      ICAP_TTRY(dense_f32(e->features, e->w_kv_cross[l], e->b_kv_cross[l], e->kv_cross[l], BL, c.D,
                          2 * HD, ICAP_ACT_NONE, s));
  }
  hipLaunchKernelGGL(tfill_i32_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, e->start_ids,
                     c.start_id, N);
  ICAP_LAUNCH_CHECK("tfill_i32_kernel");
  ICAP_HIP(hipMemsetAsync(e->lp, 0, sizeof(float) * N, s));
  ICAP_HIP(hipMemsetAsync(e->len, 0, sizeof(int32_t) * N, s));
  ICAP_HIP(hipMemsetAsync(e->fin, 0, N, s));
  // the prefix starts with the zero token of reset_state (transform_decoder.py:233-235)
  ICAP_TTRY(stack_pass(e, nullptr, 0, s));
  for (int t = 0; t < c.T; ++t) {
    const int32_t* ids = t == 0 ? e->start_ids : e->hist_ids + (size_t)(t - 1) * N;
    ICAP_TTRY(stack_pass(e, ids, t + 1, s));
    const float* lp_in = t == 0 ? e->lp : e->hist_lp + (size_t)(t - 1) * N;
    if (e->topk) {
      {
        TTimed tm(e, ICAP_TK_VOCAB, s);
        ICAP_TTRY(dense_topk_f32(e->y, e->w.fc2_kernel, e->w.fc2_bias, N, c.hidden, c.V, c.eos_id,
                                 e->topk, e->rec, e->eos_x, nullptr, s));
      }
      TTimed tm(e, ICAP_TK_BEAM, s);
      ICAP_TTRY(beam_step_from_records_f32(
          e->rec, e->eos_x, dense_topk_half_tiles(c.V), e->topk, lp_in, e->fin, e->len, t, c.eos_id,
          c.length_penalty_weight, e->hist_scores + (size_t)t * N, e->hist_ids + (size_t)t * N,
          e->hist_parents + (size_t)t * N, e->hist_lp + (size_t)t * N, e->fin, e->len, c.B, c.W, c.V,
          s));
      continue;
    }
    {
      TTimed tm(e, ICAP_TK_VOCAB, s);
      ICAP_TTRY(dense_f32(e->y, e->w.fc2_kernel, e->w.fc2_bias, e->logits, N, c.hidden, c.V,
                          ICAP_ACT_NONE, s));
    }
    TTimed tm(e, ICAP_TK_BEAM, s);
    ICAP_TTRY(beam_step_f32(e->logits, lp_in, e->fin, e->len, t, 0, c.eos_id, c.length_penalty_weight,
                            0, e->hist_scores + (size_t)t * N, e->hist_ids + (size_t)t * N,
                            e->hist_parents + (size_t)t * N, e->hist_lp + (size_t)t * N, e->fin,
                            e->len, c.B, c.W, c.V, e->beam_ws, e->beam_ws_bytes, s));
  }
  TTimed tm(e, ICAP_TK_BACKTRACK, s);
  return backtrack(e->hist_ids, e->hist_parents, e->hist_scores + (size_t)(c.T - 1) * N, e->fin, tokens,
                   nullptr, best_score, c.T, c.B, c.W, s);
}

}  // namespace
}  // namespace icap

using namespace icap;

extern "C" {

int icap_position_table_f32(float* table, int length, int hidden, icap_stream_t stream) {
  return position_table_f32(table, length, hidden, (hipStream_t)stream);
}

int icap_embed_pos_f32(const float* table, const int32_t* ids, const float* pos_row, float* x, int N,
                       int V, int D, icap_stream_t stream) {
  return embed_pos_f32(table, ids, pos_row, x, N, V, D, (hipStream_t)stream);
}

int icap_add_f32(float* x, const float* o, long long n, long long period, icap_stream_t stream) {
  return add_f32(x, o, n, period, (hipStream_t)stream);
}

int icap_residual_layernorm_f32(float* x, const float* o, const float* gamma, const float* beta,
                                float* y, int N, int D, float eps, icap_stream_t stream) {
  return residual_layernorm_f32(x, o, gamma, beta, y, N, D, eps, (hipStream_t)stream);
}

int icap_mha_self_step_f32(const float* q, const float* kv_cache, float* out, int N, int heads,
                           int key_dim, int pos, int mask_mode, icap_stream_t stream) {
  return mha_self_step_f32(q, kv_cache, out, N, heads, key_dim, pos, mask_mode, (hipStream_t)stream);
}

int icap_mha_cross_step_f32(const float* q, const float* kv, float* out, int G, int rows_per_group,
                            int row_stride_g, int row_stride_k, int L, int heads, int key_dim,
                            const float* mask, int mask_per_group, icap_stream_t stream) {
  return mha_cross_step_f32(q, kv, out, G, rows_per_group, row_stride_g, row_stride_k, L, heads,
                            key_dim, mask, mask_per_group, (hipStream_t)stream);
}

int icap_mha_scores_f32(const float* q, const float* k, const float* mask, float* probs, int Bf, int T,
                        int S, int heads, int key_dim, int softmax_segment, icap_stream_t stream) {
  return mha_scores_f32(q, k, mask, probs, Bf, T, S, heads, key_dim, softmax_segment,
                        (hipStream_t)stream);
}

int icap_mha_combine_f32(const float* probs, const float* v, float* out, int Bf, int T, int S,
                         int heads, int value_dim, icap_stream_t stream) {
  return mha_combine_f32(probs, v, out, Bf, T, S, heads, value_dim, (hipStream_t)stream);
}

int icap_tengine_refresh_weights(icap_tengine* e, icap_stream_t stream) {
  ICAP_REQUIRE(e, "icap_tengine_refresh_weights: null engine");
  const icap_tconfig& c = e->cfg;
  hipStream_t s = (hipStream_t)stream;
  const int HD = c.heads * c.key_dim;
  auto pack2 = [&](const float* a, const float* b, float* dst, int K) -> int {
    const long long n = (long long)K * HD;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(copy_columns_kernel, grid, block, 0, s, a, dst, K, HD, 2 * HD, 0);
    hipLaunchKernelGGL(copy_columns_kernel, grid, block, 0, s, b, dst, K, HD, 2 * HD, HD);
    ICAP_LAUNCH_CHECK("copy_columns_kernel");
    return ICAP_OK;
  };
  for (int l = 0; l < c.layers; ++l) {
    const icap_tlayer& L = e->layers[l];
    ICAP_TTRY(pack2(L.self_k_kernel, L.self_v_kernel, e->w_kv_self[l], c.hidden));
    ICAP_TTRY(pack2(L.self_k_bias, L.self_v_bias, e->b_kv_self[l], 1));
    ICAP_TTRY(pack2(L.cross_k_kernel, L.cross_v_kernel, e->w_kv_cross[l], c.D));
    ICAP_TTRY(pack2(L.cross_k_bias, L.cross_v_bias, e->b_kv_cross[l], 1));
  }
  return position_table_f32(e->pos_table, c.T + 1, c.hidden, s);
}

int icap_tengine_create(const icap_tconfig* config, const icap_tweights* weights, icap_tengine** out) {
  ICAP_REQUIRE(config && weights && out, "icap_tengine_create: null argument");
  *out = nullptr;
  ICAP_TTRY(check_tconfig(*config));
  ICAP_REQUIRE(weights->enc_kernel && weights->enc_bias && weights->embedding && weights->out_ln_gamma &&
                   weights->out_ln_beta && weights->fc2_kernel && weights->fc2_bias && weights->layers,
               "icap_tengine_create: null weight pointer");
  for (int l = 0; l < config->layers; ++l) {
    const void* const* lp = reinterpret_cast<const void* const*>(&weights->layers[l]);
    for (size_t i = 0; i < sizeof(icap_tlayer) / sizeof(void*); ++i)
      ICAP_REQUIRE(lp[i] != nullptr, "icap_tengine_create: layer %d weight pointer %zu is null", l, i);
  }
  icap_tengine* e = new (std::nothrow) icap_tengine();
  ICAP_REQUIRE(e, "icap_tengine_create: out of host memory");
  e->cfg = *config;
  e->w = *weights;
  e->layers.assign(weights->layers, weights->layers + config->layers);
  e->w.layers = e->layers.data();
  const icap_tconfig& c = e->cfg;
  const size_t N = (size_t)c.B * c.W, BL = (size_t)c.B * c.L, TN = (size_t)c.T * N;
  const size_t HD = (size_t)c.heads * c.key_dim, Tcap = (size_t)c.T + 1;
  {
    const int tk = c.W <= 1 ? 1 : (c.W <= 3 ? 3 : 5);
    const bool ok = c.W <= 5 &&
                    dense_topk_supported(weights->fc2_kernel, weights->fc2_kernel, (int)N, c.hidden, c.V, tk) &&
                    beam_records_supported(c.W, dense_topk_half_tiles(c.V), tk);
    e->topk = ok ? tk : 0;
  }
  TCarver cv;
  const size_t o_feat = cv.take<float>(BL * c.D), o_x = cv.take<float>(N * c.hidden),
               o_y = cv.take<float>(N * c.hidden), o_q = cv.take<float>(N * HD),
               o_att = cv.take<float>(N * HD), o_o = cv.take<float>(N * c.hidden),
               o_f = cv.take<float>(N * c.filter), o_pos = cv.take<float>(Tcap * c.hidden),
               o_logits = cv.take<float>(e->topk ? 0 : N * c.V), o_lp = cv.take<float>(N),
               o_hs = cv.take<float>(TN), o_hl = cv.take<float>(TN), o_len = cv.take<int32_t>(N),
               o_hi = cv.take<int32_t>(TN), o_hp = cv.take<int32_t>(TN), o_si = cv.take<int32_t>(N),
               o_fin = cv.take<uint8_t>(N), o_bws = cv.take<char>(beam_step_workspace_bytes(c.B, c.W)),
               o_rec = cv.take<float>(e->topk ? N * dense_topk_half_tiles(c.V) *
                                                    dense_topk_record_floats(e->topk)
                                              : 0),
               o_eosx = cv.take<float>(N);
  std::vector<size_t> o_kvs(c.layers), o_kvc(c.layers), o_wks(c.layers), o_bks(c.layers),
      o_wkc(c.layers), o_bkc(c.layers);
  for (int l = 0; l < c.layers; ++l) {
    o_kvs[l] = cv.take<float>(Tcap * N * 2 * HD);
    o_kvc[l] = cv.take<float>(BL * 2 * HD);
    o_wks[l] = cv.take<float>((size_t)c.hidden * 2 * HD);
    o_bks[l] = cv.take<float>(2 * HD);
    o_wkc[l] = cv.take<float>((size_t)c.D * 2 * HD);
    o_bkc[l] = cv.take<float>(2 * HD);
  }
  e->ws_bytes = (cv.off + 255) / 256 * 256;
  const hipError_t err = hipMalloc(reinterpret_cast<void**>(&e->ws), e->ws_bytes);
  if (err != hipSuccess) {
    const size_t need = e->ws_bytes;
    delete e;
    return fail(ICAP_ERR_HIP, "icap_tengine_create: %s (workspace of %zu bytes)", hipGetErrorString(err), need);
  }
  char* b = e->ws;
  e->features = (float*)(b + o_feat); e->x = (float*)(b + o_x); e->y = (float*)(b + o_y);
  e->q = (float*)(b + o_q); e->att = (float*)(b + o_att); e->o = (float*)(b + o_o);
  e->f = (float*)(b + o_f); e->pos_table = (float*)(b + o_pos); e->logits = (float*)(b + o_logits);
  e->lp = (float*)(b + o_lp); e->hist_scores = (float*)(b + o_hs); e->hist_lp = (float*)(b + o_hl);
  e->len = (int32_t*)(b + o_len); e->hist_ids = (int32_t*)(b + o_hi);
  e->hist_parents = (int32_t*)(b + o_hp); e->start_ids = (int32_t*)(b + o_si);
  e->fin = (uint8_t*)(b + o_fin); e->beam_ws = b + o_bws;
  e->beam_ws_bytes = beam_step_workspace_bytes(c.B, c.W);
  e->rec = (float*)(b + o_rec); e->eos_x = (float*)(b + o_eosx);
  for (int l = 0; l < c.layers; ++l) {
    e->kv_self.push_back((float*)(b + o_kvs[l]));
    e->kv_cross.push_back((float*)(b + o_kvc[l]));
    e->w_kv_self.push_back((float*)(b + o_wks[l]));
    e->b_kv_self.push_back((float*)(b + o_bks[l]));
    e->w_kv_cross.push_back((float*)(b + o_wkc[l]));
    e->b_kv_cross.push_back((float*)(b + o_bkc[l]));
  }
  int rc = icap_tengine_refresh_weights(e, nullptr);
  if (rc == ICAP_OK && hipStreamSynchronize(nullptr) != hipSuccess)
    rc = fail(ICAP_ERR_HIP, "icap_tengine_create: synchronisation failed");
  if (rc != ICAP_OK) {
    icap_tengine_destroy(e);
    return rc;
  }
  *out = e;
  return ICAP_OK;
}

int icap_tengine_destroy(icap_tengine* e) {
  if (!e) return ICAP_OK;
  for (auto& sp : e->spans) {
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  if (e->ws) (void)hipFree(e->ws);
  delete e;
  return ICAP_OK;
}

size_t icap_tengine_workspace_bytes(const icap_tengine* e) { return e ? e->ws_bytes : 0; }

int icap_tdecode(icap_tengine* e, const float* img, int32_t* tokens, float* best_score,
                 icap_stream_t stream) {
  ICAP_REQUIRE(e && img && tokens, "icap_tdecode: null argument");
  return tdecode_impl(e, img, tokens, best_score, (hipStream_t)stream);
}

int icap_tengine_trace(const icap_tengine* e, int32_t* ids, int32_t* parents, float* scores,
                       float* log_probs, icap_stream_t stream) {
  ICAP_REQUIRE(e, "icap_tengine_trace: null engine");
  const size_t n = (size_t)e->cfg.T * e->cfg.B * e->cfg.W * 4;
  hipStream_t s = (hipStream_t)stream;
  if (ids) ICAP_HIP(hipMemcpyAsync(ids, e->hist_ids, n, hipMemcpyDeviceToDevice, s));
  if (parents) ICAP_HIP(hipMemcpyAsync(parents, e->hist_parents, n, hipMemcpyDeviceToDevice, s));
  if (scores) ICAP_HIP(hipMemcpyAsync(scores, e->hist_scores, n, hipMemcpyDeviceToDevice, s));
  if (log_probs) ICAP_HIP(hipMemcpyAsync(log_probs, e->hist_lp, n, hipMemcpyDeviceToDevice, s));
  return ICAP_OK;
}

int icap_tengine_last_output(const icap_tengine* e, float* out, icap_stream_t stream) {
  ICAP_REQUIRE(e && out, "icap_tengine_last_output: null argument");
  ICAP_HIP(hipMemcpyAsync(out, e->y, sizeof(float) * e->cfg.B * e->cfg.W * e->cfg.hidden,
                          hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return ICAP_OK;
}

int icap_tengine_set_timing(icap_tengine* e, unsigned class_mask) {
  ICAP_REQUIRE(e, "icap_tengine_set_timing: null engine");
  if (class_mask) {
    // per decode: T + 1 passes of the stack (eleven launches per layer, three around it) plus
    // the once-per-decode classes; the pool serves ICAP_TTIMING_POOL_DECODES decodes
    const size_t per_decode =
        8 + ((size_t)e->cfg.T + 1) * (11 * (size_t)e->cfg.layers + 5);
    const size_t want = (size_t)ICAP_TTIMING_POOL_DECODES * per_decode;
    while (e->spans.size() < want) {
      icap_tengine::Span sp;
      sp.cls = 0;
      sp.a = sp.b = nullptr;
      if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) {
        if (sp.a) (void)hipEventDestroy(sp.a);
        return fail(ICAP_ERR_HIP, "icap_tengine_set_timing: cannot create the timing events");
      }
      e->spans.push_back(sp);
    }
  }
  e->timing = class_mask;
  e->spans_used = 0;
  return ICAP_OK;
}

int icap_tengine_kernel_times(icap_tengine* e, float* ms, int32_t* launches) {
  ICAP_REQUIRE(e && ms && launches, "icap_tengine_kernel_times: null argument");
  for (int i = 0; i < ICAP_TK_COUNT; ++i) {
    ms[i] = 0.f;
    launches[i] = 0;
  }
  for (size_t i = 0; i < e->spans_used; ++i) {
    auto& sp = e->spans[i];
    ICAP_HIP(hipEventSynchronize(sp.b));
    float t = 0.f;
    ICAP_HIP(hipEventElapsedTime(&t, sp.a, sp.b));
    ms[sp.cls] += t;
    launches[sp.cls] += 1;
  }
  return ICAP_OK;
}

}  // extern "C"
