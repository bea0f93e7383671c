// C ABI of libicap: argument checking wrappers around the kernels plus the fused
// whole-sequence decode engine (workspace, step loop, optional per-kernel timing).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "icap_common.h"
#include "topk_epilogue.h"

namespace icap {

std::string& last_error() {
  static thread_local std::string msg;
  return msg;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

// kernels (gemm_f32.hip, gemm_bf16.hip, gemm_split.hip, attention.hip, beam.hip, beam_unfused.hip)
int dense_f32(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
int lstm_step_f32(const float*, const float*, const float*, const int32_t*, const float*,
                  const float*, const float*, const float*, const float*, const float*, float*,
                  float*, int, int, int, int, int, hipStream_t);
int gru_pack_f32(const float*, const float*, const float*, float*, float*, float*, int, int,
                 hipStream_t);
int gru_step_f32(const float*, const float*, const float*, const int32_t*, const float*,
                 const float*, const float*, const float*, const float*, float*, int, int, int, int,
                 int, hipStream_t);
size_t packed_weight_bytes(int K, int N);
int pack_weight_bf16(const float*, void*, int, int, int, int, int, hipStream_t);
int dense_bf16(const float*, const void*, const float*, float*, int, int, int, int, hipStream_t);
int lstm_step_bf16(const float*, const float*, const float*, const void*, const float*, const float*,
                   const int32_t*, float*, float*, int, int, int, int, hipStream_t);
int gru_step_bf16(const float*, const float*, const void*, const float*, const float*, const int32_t*,
                  float*, int, int, int, int, hipStream_t);
struct SmallLoopArgs;
bool small_loop_supported(int B, int W, int C, int H, int U, int L, int D, int V);
int small_loop_decode(const icap_config& c, const icap_weights& w, const float* wout, const float* bout,
                      const float* emb_proj, float att_v_bias, float fc3_bias, const float* pf,
                      const float* epf, const int* pf_extreme, const float* features, float* h0, float* h1,
                      float* cst, float* q, float* ctx, float* logits, uint8_t* fin, int32_t* len,
                      int32_t* hist_ids, int32_t* hist_parents, float* hist_scores, float* hist_lp,
                      unsigned* bar, int* err, int compute_units, hipStream_t s);
int cell_gates_bf16(const float*, const float*, const void*, float*, int, int, int, hipStream_t);
int cell_update_f32(int, const float*, const float*, const float*, const int32_t*, const float*, float*,
                    float*, int, int, int, hipStream_t);
int dense_topk_half_tiles(int N);
int dense_topk_record_floats(int top_k);
bool dense_topk_supported(const float*, const float*, int, int, int, int);
int dense_topk_f32(const float*, const float*, const float*, int, int, int, int, int, float*, float*,
                   const LogitMasks*, hipStream_t);
int logit_masks_f32(float*, int, int, const LogitMasks&, hipStream_t);
int expand_rows_i32(const int32_t*, int32_t*, int, int, int, hipStream_t);
int sample_categorical_f32(const float*, const float*, const int32_t*, int, int, int, int, int32_t*, int,
                           int, hipStream_t);
bool beam_records_supported(int W, int half_tiles, int top_k);
size_t packed_split_bytes(int K, int N);
int pack_weight_split(const float*, void*, int, int, int, int, int, hipStream_t);
int dense_split(const float*, const void*, const float*, float*, int, int, int, int, hipStream_t);
bool dense_topk_split_supported(const float*, int, int, int, int);
int dense_topk_split(const float*, const void*, const float*, int, int, int, int, int, float*,
                     float*, const LogitMasks*, hipStream_t);
int lstm_step_split(const float*, const float*, const float*, const void*, const float*,
                    const float*, const int32_t*, float*, float*, int, int, int, int, hipStream_t);
bool dense_topk_bf16_supported(const float*, int, int, int, int);
int dense_topk_bf16(const float*, const void*, const float*, int, int, int, int, int, float*, float*,
                    const LogitMasks*, hipStream_t);
int dense_bf16_dual(const float*, const void*, const float*, float*, void*, int, int, int, int, int,
                    hipStream_t);
bool dense_topk_bf16_a16_blocked(int, int, int, int);
int dense_topk_bf16_a16(const void*, int, const void*, const float*, int, int, int, int, int, float*,
                        float*, const LogitMasks*, hipStream_t);
int beam_step_from_records_f32(const float*, const float*, int, int, const float*, const uint8_t*,
                               const int32_t*, int, int, double, float*, int32_t*, int32_t*, float*,
                               uint8_t*, int32_t*, int, int, int, hipStream_t, int compact_first = 0);
int dense_f32_strided(const float*, int, const float*, const float*, float*, int, int, int, int,
                      hipStream_t);
int dense_topk_f32_strided(const float*, int, const float*, const float*, int, int, int, int, int,
                           float*, float*, const LogitMasks*, hipStream_t);
bool dense_topk_attention_supported(int R, int L, int U, int D, int H, int top_k);
bool dense_topk_beam_supported(int M, int W, int half_tiles, int top_k, int time, int compact_first);
int dense_topk_beam_f32(const float*, int, const float*, const float*, int, int, int, int, int, float*,
                        float*, const float*, const uint8_t*, const int32_t*, int, double, float*,
                        int32_t*, int32_t*, float*, uint8_t*, int32_t*, int, int, int, unsigned*,
                        unsigned, int*, hipStream_t);
int backtrack_guarded(const int32_t*, const int32_t*, const float*, const uint8_t*, int32_t*, int32_t*,
                      float*, int, int, int, const int*, hipStream_t);
int dense_topk_attention_f32(const float*, int, const float*, const float*, int, int, int, int, int,
                             float*, float*, const LogitMasks*, const float*, const float*,
                             const float*, float, const float*, const float*, float, const float*,
                             const float*, float*, int, int, int, int, int, int, int, int, int,
                             hipStream_t);
bool dense_topk_attq_supported(int R, int L, int U, int D, int H, int top_k);
bool dense_topk_bf16_attq_supported(int R, int L, int U, int D, int H, int top_k);
int dense_topk_bf16_attq_tiles(int rows, int U);
int dense_topk_bf16_attq(const void*, const void*, const float*, int, int, int, int, int, float*, float*,
                         const LogitMasks*, const float*, const float*, const float*, float,
                         const float*, const float*, float, const void*, const float*, float*, float*,
                         int, int, int, int, int, int, int, int, int, unsigned*, unsigned, int*,
                         hipStream_t, const float* epf = nullptr, const int* pf_extreme = nullptr);
int dense_topk_attq_tiles(int rows, int U);
int dense_topk_attq_f32(const float*, int, const float*, const float*, int, int, int, int, int,
                        float*, float*, const LogitMasks*, const float*, const float*,
                        const float*, float, const float*, const float*, float, const float*,
                        const float*, float*, float*, int, int, int, int, int, int, int, int, int,
                        unsigned*, unsigned, int*, hipStream_t, const float* epf = nullptr,
                        const int* pf_extreme = nullptr);
bool attention_fq_supported(int R, int L, int U, int D, int H);
int attention_fq_f32(const float*, const float*, const float*, float, const float*, const float*,
                     float, const float*, const float*, float*, int, int, int, int, int, int, int,
                     int, int, hipStream_t);
int attention_act_f32(const float*, const float*, const float*, const float*, float, const float*,
                      const float*, const float*, float, float*, float*, int, int, int, int, int, int,
                      int, int, int, int, hipStream_t, const float* epf = nullptr,
                      const int* pf_extreme = nullptr, const void* epf16 = nullptr,
                      const void* features16 = nullptr);
int attention_f32(const float*, const float*, const float*, const float*, float, const float*,
                  const float*, const float*, float, float*, float*, int, int, int, int, int, int,
                  int, int, int, hipStream_t, const float* epf = nullptr,
                  const int* pf_extreme = nullptr, const void* epf16 = nullptr,
                  const void* features16 = nullptr);
int attention_exp_table_f32(const float* pf, float* epf, int* extreme, size_t n, hipStream_t s);
int attention_exp_table_bf16(const float* pf, void* epf16, int* extreme, size_t n, hipStream_t s);
int to_bf16(const float* x, void* y16, size_t n, hipStream_t s);
int pool_features_f32(const float*, const float*, float*, int, int, int, hipStream_t);
int expand_rows_f32(const float*, float*, int, int, int, float, int, hipStream_t);
int embedding_f32(const float*, const int32_t*, float*, int, int, int, hipStream_t);
int beam_step_f32(const float*, const float*, const uint8_t*, const int32_t*, int, int, int,
                  double, int, float*, int32_t*, int32_t*, float*, uint8_t*, int32_t*, int, int,
                  int, void*, size_t, hipStream_t);
int beam_step_rows_f32(const float*, const float*, const uint8_t*, const int32_t*, int, int, int,
                       double, int, float*, int32_t*, int32_t*, float*, uint8_t*, int32_t*, int, int,
                       int, int, void*, size_t, hipStream_t, const int* run_flag = nullptr,
                       int* clear_flag = nullptr);
int beam_step_from_records_wide_f32(const float*, const float*, int, int, const float*, const uint8_t*,
                                    const int32_t*, int, int, double, float*, int32_t*, int32_t*,
                                    float*, uint8_t*, int32_t*, int, int, int, int*, hipStream_t,
                                    int compact_first = 0);
bool dense_f32_flagged_supported(const float*, const float*, int, int, int, int);
int dense_f32_flagged(const float*, int, const float*, const float*, float*, int, int, int, const int*,
                      hipStream_t);
size_t beam_step_workspace_bytes(int B, int W);
int beam_step_static_f32(const float*, const float*, const uint8_t*, const int32_t*, int, int, int,
                         double, int, float*, int32_t*, int32_t*, float*, uint8_t*, int32_t*, int, int,
                         int, hipStream_t);
int beam_scores_f32(const float*, const float*, const uint8_t*, const int32_t*, int, double, int,
                    float*, float*, int, int, int, hipStream_t);
int beam_gather_f32(const float*, const uint8_t*, const int32_t*, const int32_t*, int, int32_t*,
                    int32_t*, float*, uint8_t*, int32_t*, int, int, int, int, hipStream_t);
int mask_probs_f32(const float*, const uint8_t*, float*, int, int, int, hipStream_t);
int hyp_score_f32(const float*, const int32_t*, double, float*, long long, hipStream_t);
int batch_gather_f32(const float*, const int32_t*, float*, int, int, int, hipStream_t);
int backtrack(const int32_t*, const int32_t*, const float*, const uint8_t*, int32_t*, int32_t*,
              float*, int, int, int, hipStream_t);

namespace {
__global__ void fill_i32_kernel(int32_t* p, int32_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// start tokens and the initial beam state (zeros, beam_search.py:85-98) in one launch and one
// graph node instead of four memset calls.  (An earlier form with memset calls returned stale
// state on replays of a captured decode after the first; a stand-alone graph of memset nodes
// in front of a kernel, tools/lab/graph_memset_lab.hip, runs them on every replay, so that
// symptom stays unexplained - the kernel form has no such node.)
__global__ void init_beam_state_kernel(int32_t* start_ids, int32_t start_id, float* lp, int32_t* len,
                                       uint8_t* fin, int n, unsigned* words2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4) words2[i] = 0u;  // the query-tile counter and its error word; the one-launch loop's barrier count and error word
  if (i < n) {
    start_ids[i] = start_id;
    lp[i] = 0.f;
    len[i] = 0;
    fin[i] = 0;
  }
}
}  // namespace

}  // namespace icap

namespace icap {
// ONE side stream per device for every engine of the process, created with the first engine on
// that device and kept until the process ends.  A stream of a priority class of its own costs
// the process a hardware queue of its own, and the HIP runtime shares hardware queues between
// streams only within a priority class, GPU_MAX_HW_QUEUES (4) per class: an engine-owned
// stream per engine made the fifth hardware queue of a process as soon as two engines were
// alive beside three streams of the caller.  On this runtime (ROCm 7.0 / 7.2, reproduced
// without this library by tools/lab/null_stream_forkjoin_lab.hip) a decode whose legacy
// default (null) stream and fifth queue wait on each other's events never finishes: the
// queues stop consuming packets although every wait has its signal queued.  With one shared
// stream the library adds one queue per device whatever the number of engines.  Engines that
// decode at the same time on different streams interleave their side work on it; every piece
// waits only for an event recorded before it was enqueued, so the interleaving is ordered
// correctly and cannot deadlock, it merely serialises the side work of concurrent engines.
hipError_t shared_side_stream(int device, hipStream_t* out) {
  constexpr int MAX_DEVICES = 64;
  static std::mutex mu;
  static hipStream_t streams[MAX_DEVICES] = {};
  if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!streams[device]) {
    // The side stream carries work that is off the critical path: the lowest priority lets
    // its workgroups fill the compute units the main stream leaves idle (small launches, kernel
    // tails) instead of competing with the vocab projection.  ICAP_SIDE_PRIORITY=default|high
    // (read once per device) selects another class.
    int prio_low = 0, prio_high = 0;
    hipError_t err = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (err != hipSuccess) return err;
    const char* prio_env = getenv("ICAP_SIDE_PRIORITY");
    if (prio_env && prio_env[0] == 'd')
      err = hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
    else
      err = hipStreamCreateWithPriority(&streams[device], hipStreamNonBlocking,
                                        prio_env && prio_env[0] == 'h' ? prio_high : prio_low);
    if (err != hipSuccess) {
      streams[device] = nullptr;
      return err;
    }
  }
  *out = streams[device];
  return hipSuccess;
}
}  // namespace icap

struct icap_engine {
  icap_config cfg;
  icap_weights w;
  int device = 0;
  float att_v_bias = 0.f, fc3_bias = 0.f;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  // workspace views
  float *features, *pf, *pooled, *init_h, *init_c;
  float *h[2], *c, *q, *ctx, *fc1o, *logits;
  uint16_t* fc1o16 = nullptr;  // bf16 mode: fc1's output rounded to bf16, the projection's A operand
  float *lp, *hist_scores, *hist_lp;
  int32_t *len, *hist_ids, *hist_parents, *start_ids;
  int32_t* row_len = nullptr;  // [N] target length of every beam row (baseline searcher)
  uint8_t* fin;
  char* beam_ws;
  size_t beam_ws_bytes;
  float* emb_proj;  // [V,4H] = embedding . lstm_kernel[D:D+E]  (token-only part of the gates)
  // GRU cell: the Keras variables re-laid as four pre-activation columns per unit
  float *gru_k4 = nullptr, *gru_r4 = nullptr, *gru_b4 = nullptr;
  // the gate operands the step kernels read (the caller's LSTM variables or the GRU re-layout)
  const float *cell_kernel = nullptr, *cell_recurrent = nullptr, *cell_bias = nullptr;
  // bf16 mode: weights packed as bf16 [N_pad][K] (null in fp32 mode)
  enum { PK_ENC, PK_W1, PK_INIT_H, PK_INIT_C, PK_W2, PK_LSTM, PK_LSTM_EMB, PK_FC1, PK_FC2, PK_COUNT };
  void* pk[PK_COUNT] = {};
  void* pk_fold = nullptr;  // split mode: the folded fc1 . fc2 kernel, packed like pk[PK_FC2]
  // fused vocab projection + beam step: per-row top-k records instead of the [N,V] logits
  int topk = 0;       // 0: unfused (logits buffer + beam_step kernels)
  int topk_able = 0;  // the top_k the fused path would use for this shape (0: not served)
  float *rec = nullptr, *eos_x = nullptr;
  // Beams wider than the records are deep (W > 5 on records of 5 logits per half tile): the
  // beam kernel proves per image that no unrecorded logit matters and raises a device flag
  // otherwise; the unfused projection + beam step are enqueued behind it and do nothing
  // unless the flag is up.  The step then needs the state it started from: fin / len ping-pong.
  bool wide = false;
  int* flags = nullptr;        // two words (step parity), in a 256-byte block of their own
  // exact-f32 fused path: the beam step rides in the projection's launch (dense_topk_beam_f32)
  bool beam_in_launch = false;
  unsigned* tile_rows_done = nullptr;  // two runs of counters (time-0 launch, the others) + error word
  int tile_rows = 0;
  uint8_t* fin2 = nullptr;
  int32_t* len2 = nullptr;
  long wide_rescues_forced = 0;  // test hook: raise the flag at every step
  bool a16 = true;               // bf16 mode: bf16 hand-over between fc1 and the projection (ICAP_BF16_A16=0: off)
  // bf16 mode, side-stream schedule: the cell step as two launches (cell_gates_bf16 on the side
  // stream behind the attention, cell_update_f32 on the caller's stream behind the beam step)
  // greedy decodes of a few dozen rows: the whole step loop in one launch of resident workgroups
  // (csrc/persistent.hip).  Opt-in, ICAP_SMALL_LOOP=1 at creation: measured at 32 rows its
  // three grid barriers per step (4 to 9 us each across 256 workgroups) cost more than the
  // launches they replace (78 us per step against 48).
  bool small_loop = false;
  int compute_units = 0;
  bool split_cell = false;
  float* zgates = nullptr;       // [N][4H] gate pre-activations of the next step
  // fc1 and fc2 have no activation between them (decoder/lstm_recurrent_decoder.py:41-42,
  // 106-112; the dropout is the identity at inference), so logits = h . (W1 . W2) + (b1 . W2 + b2):
  // the product of the two kernels and the combined bias are formed once per engine (sums in
  // double precision, rounded to f32 once) and the step projects the LSTM output directly -
  // one launch per step less.  fp32 and split mode; ICAP_FOLD_FC=0 / icap_engine_set_fold_fc
  // keep the two-step form.
  float *w12 = nullptr, *b12 = nullptr;
  bool fold = false;
  // side stream (the device's shared one, see shared_side_stream): the next step's attention
  // runs beside the current step's vocab/beam work
  hipStream_t side = nullptr;
  hipEvent_t ev_lstm = nullptr, ev_att = nullptr, ev_gate = nullptr;
  bool overlap = true;  // false: everything on the caller's stream (clean per-kernel timing)
  int side_att_wgs = 0;  // grid cap of the attention kernel when it runs on the side stream
  // exact-f32 fused path: the next step's query + attention ride in the first workgroups of
  // the vocab projection's launch (one stream, no events) instead of on the side stream
  bool merged = false;
  // ... with the query projection as matrix tiles of that launch (dense_topk_attq_f32): a
  // counter of finished query tiles (zeroed per decode) and an error word behind it
  bool qtiles = false;
  unsigned* q_done = nullptr;
  // exp(2 * PF) [B,L,U] and its "too large" flag: the step-invariant factor of the tanh scores
  // (null: the attention kernels take the plain form; ICAP_ATT_TABLE=0)
  float* epf = nullptr;
  int* pf_extreme = nullptr;
  // bf16 mode on the side-stream schedule, opt-in (ICAP_ATT_BF16=1): that table and the features
  // as bf16 instead - the attention kernel's two streams at half the bytes.  Off by default:
  // the kernel alone gets shorter, the decode does not (192.9k captions/s either way at B = 512,
  // beam 5; the attention is hidden beside the projection), and the scores then differ from the
  // float32-table form in the last bf16 digit.
  uint16_t* epf16 = nullptr;
  uint16_t* feat16 = nullptr;
  // The device word that the last enqueued decode's in-launch hand-over raises when one of its
  // bounded waits ran out (null: that decode had no such wait); icap_engine_decode_status reads it
  int* last_err = nullptr;
  bool force_timeout = false;  // test hook (ICAP_MERGED_FORCE_TIMEOUT=1): the waits can never be met
  // timing
  unsigned timing = 0;  // bit k set: launches of kernel class k are bracketed by events
  unsigned timing_mask = 0;  // the caller's selection; `timing` is it or 0, per decode call
  int timing_every = 1;      // bracket every n-th decode call
  long decode_calls = 0;     // since the mask was set
  struct Span { int cls; hipEvent_t a, b; };
  std::vector<Span> spans;   // event pool, reserved by icap_engine_set_timing
  size_t spans_used = 0;
  size_t spans_dropped = 0;  // launches that found the pool used up
};

namespace icap {
namespace {

// w12[k][v] = sum_j w1[k][j] * w2[j][v], b12[v] = sum_j b1[j] * w2[j][v] + b2[v] (row k == H),
// accumulated in double precision.  One thread per (k, v); a workgroup covers 256 columns of
// one row, so w1[k][j] is a broadcast and w2[j][v] a coalesced read.
__global__ __launch_bounds__(256) void fold_fc_kernel(const float* __restrict__ w1,
                                                      const float* __restrict__ b1,
                                                      const float* __restrict__ w2,
                                                      const float* __restrict__ b2,
                                                      float* __restrict__ w12,
                                                      float* __restrict__ b12, int H, int V) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;  // k == H: the bias row
  if (v >= V) return;
  const float* left = k < H ? w1 + (size_t)k * H : b1;
  double acc = 0.0;
  for (int j = 0; j < H; ++j) acc += (double)left[j] * (double)w2[(size_t)j * V + v];
  if (k < H) w12[(size_t)k * V + v] = (float)acc;
  else b12[v] = (float)(acc + (double)b2[v]);
}

struct Carver {
  size_t off = 0;
  template <typename T>
  size_t take(size_t n) {
    off = (off + 255) / 256 * 256;
    const size_t o = off;
    off += n * sizeof(T);
    return o;
  }
};

int check_config(const icap_config& c) {
  ICAP_REQUIRE(c.B > 0 && c.W > 0 && c.L > 0 && c.F > 0 && c.D > 0 && c.E > 0 && c.H > 0 &&
                   c.U > 0 && c.V > 0 && c.T > 0,
               "engine: every dimension must be positive");
  ICAP_REQUIRE(c.W <= ICAP_MAX_BEAM_WIDTH, "engine: beam width %d exceeds %d", c.W,
               ICAP_MAX_BEAM_WIDTH);
  ICAP_REQUIRE(c.W <= c.V, "engine: beam width exceeds vocab size");
  ICAP_REQUIRE(c.start_id >= 0 && c.start_id < c.V, "engine: start_id outside the vocabulary");
  ICAP_REQUIRE(c.tile_mode == ICAP_TILE_REFERENCE || c.tile_mode == ICAP_TILE_REPEAT,
               "engine: unknown tile_mode %d", c.tile_mode);
  ICAP_REQUIRE(c.U % 4 == 0 && c.D % 4 == 0, "engine: U and D must be multiples of 4");
  ICAP_REQUIRE(c.cell == ICAP_CELL_LSTM || c.cell == ICAP_CELL_GRU, "engine: unknown cell %d", c.cell);
  if (c.precision != ICAP_PRECISION_FP32 && c.precision != ICAP_PRECISION_BF16 &&
      c.precision != ICAP_PRECISION_F32_SPLIT)
    return fail(ICAP_ERR_UNSUPPORTED, "engine: precision %d is not built", c.precision);
  if (c.precision == ICAP_PRECISION_F32_SPLIT && (c.F % 32 || c.D % 32 || c.E % 32 || c.H % 32))
    return fail(ICAP_ERR_UNSUPPORTED,
                "engine: split mode needs F, D, E and H to be multiples of 32 (got %d %d %d %d)",
                c.F, c.D, c.E, c.H);
  if (c.precision == ICAP_PRECISION_BF16 &&
      (c.F % 64 || c.D % 64 || c.E % 64 || c.H % 64))
    return fail(ICAP_ERR_UNSUPPORTED,
                "engine: bf16 mode needs F, D, E and H to be multiples of 64 (got %d %d %d %d)", c.F,
                c.D, c.E, c.H);
  ICAP_REQUIRE((long long)c.B * c.W * (long long)c.V < (1LL << 40), "engine: logits too large");
  return ICAP_OK;
}

// Brackets one launch with a pair of events from the pool that icap_engine_set_timing
// reserved; a decode never creates events (it may run under graph capture and promises not to
// allocate).  When the pool is used up, later launches go unbracketed and are counted as dropped.
struct Timed {
  icap_engine* e;
  hipStream_t s;
  bool on;
  size_t idx = 0;
  Timed(icap_engine* e_, int cls, hipStream_t s_) : e(e_), s(s_), on(((e_->timing >> cls) & 1u) != 0) {
    if (!on) return;
    if (e->spans_used == e->spans.size()) {
      e->spans_dropped++;
      on = false;
      return;
    }
    idx = e->spans_used++;
    e->spans[idx].cls = cls;
    (void)hipEventRecord(e->spans[idx].a, s);
  }
  ~Timed() {
    if (on) (void)hipEventRecord(e->spans[idx].b, s);
  }
};

#define ICAP_TRY(expr)            \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != ICAP_OK) return rc_; \
  } while (0)

// y = act(x . W + b) in the engine's precision; `which` names the packed copy of `wf`.
// packed: the packed form of wf when it is not e->pk[which] (the folded output kernel)
int engine_dense(const icap_engine* e, int which, const float* x, const float* wf, const float* bias,
                 float* y, int M, int K, int N, int act, hipStream_t s, const void* packed = nullptr) {
  const void* pkw = packed ? packed : e->pk[which];
  if (e->cfg.precision == ICAP_PRECISION_BF16)
    return dense_bf16(x, pkw, bias, y, M, K, N, act, s);
  // split mode pays on the large products; the small projections (a few hundred 64x64
  // tiles) are latency-bound and stay on the exact-f32 kernels
  if (e->cfg.precision == ICAP_PRECISION_F32_SPLIT && (long long)ceil_div(M, 128) * ceil_div(N, 128) >= 256)
    return dense_split(x, pkw, bias, y, M, K, N, act, s);
  return dense_f32(x, wf, bias, y, M, K, N, act, s);
}

// The loop variants: the decode loop of recurrent_example.py:477-562 (all defaults) and the
// baseline beam searcher of utils/agent_utils.py:152-262, which starts from encoded features,
// counts beam-search time from 1 (so the first step already ranks all W rows), masks the
// start token at every step and, with target lengths, forces eos / pad by position.
struct LoopVariant {
  const float* features = nullptr;  // encoded [B,L,D]: skips the encoder
  const int32_t* seq_len = nullptr; // [B] target lengths (null: no forced eos / pad)
  int time_offset = 0;
  int mask_start = 0;
  int pad = 0;
};

int decode_impl(icap_engine* e, const float* img, int32_t* tokens, float* best_score,
                hipStream_t s, const LoopVariant& lv = LoopVariant()) {
  const icap_config& c = e->cfg;
  const icap_weights& w = e->w;
  const int N = c.B * c.W;
  const int BL = c.B * c.L;
  const bool gru = c.cell == ICAP_CELL_GRU;
  e->timing = (e->decode_calls++ % e->timing_every == 0) ? e->timing_mask : 0u;

  // the wide fused step has no masked form: the searcher's loop takes the unfused kernels there
  const int topk = (e->wide && (lv.mask_start || lv.seq_len)) ? 0 : e->topk;
  const bool wide = topk != 0 && e->wide;
  uint8_t* fin = e->fin;  // beam state of the current step (the wide step ping-pongs it)
  int32_t* len = e->len;
  const float* features = lv.features ? lv.features : e->features;
  if (!lv.features) {  // K0: features = relu(img . Wenc + benc)
    Timed t(e, ICAP_K_ENCODE, s);
    ICAP_TRY(engine_dense(e, icap_engine::PK_ENC, img, w.enc_kernel, w.enc_bias, e->features, BL,
                          c.F, c.D, ICAP_ACT_RELU, s));
  }
  {  // K1: PF = features . W1 + b1   (step invariant, per image)
    Timed t(e, ICAP_K_FEAT_PROJ, s);
    ICAP_TRY(engine_dense(e, icap_engine::PK_W1, features, w.att_w1_kernel, w.att_w1_bias, e->pf,
                          BL, c.D, c.U, ICAP_ACT_NONE, s));
  }
  if (e->epf || e->epf16) {  // exp(2 PF) for the factored tanh scores of every step's attention
    Timed t(e, ICAP_K_FEAT_PROJ, s);
    hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, s, e->pf_extreme, 0, 1);
    ICAP_LAUNCH_CHECK("fill_i32_kernel");
    if (e->epf) ICAP_TRY(attention_exp_table_f32(e->pf, e->epf, e->pf_extreme, (size_t)BL * c.U, s));
    if (e->epf16) {
      ICAP_TRY(attention_exp_table_bf16(e->pf, e->epf16, e->pf_extreme, (size_t)BL * c.U, s));
      ICAP_TRY(to_bf16(features, e->feat16, (size_t)BL * c.D, s));
    }
  }
  LogitMasks masks;
  if (lv.mask_start) masks.start_col = c.start_id;
  if (lv.seq_len) {
    ICAP_TRY(expand_rows_i32(lv.seq_len, e->row_len, c.B, c.W, c.tile_mode, s));
    masks.row_len = e->row_len;
    masks.eos = c.eos_id;
    masks.pad = lv.pad;
  }
  {  // K2: h0, c0 per image, expanded to the beam rows
    Timed t(e, ICAP_K_INIT_STATE, s);
    ICAP_TRY(pool_features_f32(features, nullptr, e->pooled, c.B, c.L, c.D, s));
    ICAP_TRY(engine_dense(e, icap_engine::PK_INIT_H, e->pooled, w.init_h_kernel, w.init_h_bias,
                          e->init_h, c.B, c.D, c.H, ICAP_ACT_NONE, s));
    ICAP_TRY(expand_rows_f32(e->init_h, e->h[0], c.B, c.W, c.H, c.reset_state_from_feat,
                             c.tile_mode, s));
    if (!gru) {
      ICAP_TRY(engine_dense(e, icap_engine::PK_INIT_C, e->pooled, w.init_c_kernel, w.init_c_bias,
                            e->init_c, c.B, c.D, c.H, ICAP_ACT_NONE, s));
      ICAP_TRY(expand_rows_f32(e->init_c, e->c, c.B, c.W, c.H, c.reset_state_from_feat,
                               c.tile_mode, s));
    }
    hipLaunchKernelGGL(init_beam_state_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, e->start_ids,
                       c.start_id, e->lp, e->len, e->fin, N, e->q_done);
    ICAP_LAUNCH_CHECK("init_beam_state_kernel");
    if (wide) {
      hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, s, e->flags, 0, 64);
      ICAP_LAUNCH_CHECK("fill_i32_kernel");
    }
  }
  // The beam step inside the projection's launch: exact-f32 fused path without logit masks
  const bool beam_in_launch = e->beam_in_launch && topk != 0 && !wide && !masks.any() &&
                              !(e->merged && e->overlap);
  unsigned* done0 = e->tile_rows_done;
  unsigned* done1 = e->tile_rows_done + e->tile_rows;
  int* beam_err = reinterpret_cast<int*>(e->tile_rows_done + 2 * e->tile_rows);
  unsigned launches1 = 0;
  if (beam_in_launch) {
    const int words = 2 * e->tile_rows + 1;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(ceil_div(words, 256)), dim3(256), 0, s,
                       reinterpret_cast<int32_t*>(e->tile_rows_done), 0, words);
    ICAP_LAUNCH_CHECK("fill_i32_kernel");
  }

  const int stride_g = c.tile_mode == ICAP_TILE_REPEAT ? c.W : 1;
  const int stride_k = c.tile_mode == ICAP_TILE_REPEAT ? 1 : c.B;
  // The recurrent state is not reordered by parent beam (recurrent_example.py:526-529),
  // so the attention of step t+1 depends only on the LSTM of step t, not on the vocab
  // projection / beam step of step t.  It runs on the side stream beside them:
  //   main: lstm_t -> fc1_t -> vocab_t -> beam_t -> lstm_{t+1} ...
  //   side:        \-> query_{t+1} -> attention_{t+1} -/
  // state: the attention query and the beta gate read the LSTM cell state c
  // (lstm_recurrent_decoder.py:80) or the GRU state h (recurrent_decoder.py:77,90)
  // Opt-in (ICAP_SIDE_FQ=1): on the side stream in exact-f32 mode the query projection is
  // folded into the attention launch (attention_fq_f32), which keeps the query kernel from
  // competing with fc1 for whole compute units - but the longer attention costs the vocab
  // projection more than that saves (90k against 97.5k captions/s at the headline shape).
  const char* fq_env = getenv("ICAP_SIDE_FQ");
  const bool side_fq = c.precision == ICAP_PRECISION_FP32 && fq_env && fq_env[0] == '1' &&
                       attention_fq_supported(c.W, c.L, c.U, c.D, c.H) &&
                       aligned16(w.att_v_kernel) && aligned16(w.att_w2_kernel);
  // The side stream's attention has to be resident before the vocab projection's workgroups
  // arrive: launched after them it waits for the projection's first round of tiles and ends
  // up on the critical path.  Its launch (behind the query kernel) and the projection's
  // (behind fc1) are a close race; one idle event record in front of the fused projection
  // (a few microseconds on the main stream) settles it where it matters: split mode (113k
  // against 104k captions/s at the headline shape) and long exact-f32 projections (B = 1024,
  // beam 3: 106.7k against 100.8k; B = 512, beam 5: 57.0k against 56.7k).  At the headline
  // shape itself the exact-f32 path is better without it (99.8k against 98.8k; B = 384: 84.9k
  // against 84.1k), the short bf16 projection loses with it (187k against 203k at beam 3),
  // and so does the plain projection of small batches (1-2 %).  Two or three records lose
  // the gain again; a timing bracket around the projection is such a record already.
  // Tuning knob ICAP_VOCAB_GATE=0|1.
  static const char* gate_env = getenv("ICAP_VOCAB_GATE");
  const long gate_tiles = (long)ceil_div(N, 128) * ceil_div(c.V, 128);
  const bool vocab_gate = gate_env ? gate_env[0] == '1'
                                   : (c.precision == ICAP_PRECISION_F32_SPLIT ||
                                      (c.precision == ICAP_PRECISION_FP32 && gate_tiles >= 1400));
  auto attention_step = [&](hipStream_t st, const float* state) -> int {
    if (st != s && side_fq) {  // K4+K5 in one launch
      Timed tm(e, ICAP_K_ATTENTION, st);
      return attention_fq_f32(e->pf, features, w.att_v_kernel, e->att_v_bias, state, w.fc3_kernel,
                              e->fc3_bias, w.att_w2_kernel, w.att_w2_bias, e->ctx, c.B, c.W,
                              stride_g, stride_k, c.L, c.U, c.D, c.H, e->side_att_wgs, st);
    }
    {  // K4: q = W2(state) + b2
      Timed tm(e, ICAP_K_ATTN_QUERY, st);
      ICAP_TRY(engine_dense(e, icap_engine::PK_W2, state, w.att_w2_kernel, w.att_w2_bias, e->q, N,
                            c.H, c.U, ICAP_ACT_NONE, st));
    }
    {  // K5: additive attention + beta gate
      Timed tm(e, ICAP_K_ATTENTION, st);
      ICAP_TRY(attention_f32(e->pf, features, e->q, w.att_v_kernel, e->att_v_bias, nullptr,
                             state, w.fc3_kernel, e->fc3_bias, e->ctx, nullptr, c.B, c.W, stride_g,
                             stride_k, c.L, c.U, c.D, c.H, st == s ? 0 : e->side_att_wgs, st, e->epf,
                             e->pf_extreme, e->epf16, e->feat16));
    }
    return ICAP_OK;
  };
  // Exact-f32 fused path: query_{t+1} + attention_{t+1} are the first workgroups of the
  // vocab_t launch instead (dense_topk_attention_f32), everything on the caller's stream:
  //   main: lstm_t -> fc1_t -> [attention_{t+1} | vocab_t] -> beam_t -> lstm_{t+1} ...
  // (the same schedule whether or not the stream is being captured into a graph)
  const bool bf16 = c.precision == ICAP_PRECISION_BF16;
  // (bf16 mode: the merged launch takes the row-major bf16 activations fc1 hands over)
  const bool merged = e->merged && e->overlap && topk != 0 &&
                      (!bf16 || (e->a16 && !dense_topk_bf16_a16_blocked(N, c.H, c.V, topk)));
  const bool side = e->overlap && !merged;
  const int att_blocks = e->side_att_wgs > 0 ? e->side_att_wgs : c.B;
  // the output projection's operands: the folded kernel applied to the LSTM output, or fc2
  // applied to fc1's output
  const bool fold = e->fold && e->w12 != nullptr;
  const float* vw = fold ? e->w12 : w.fc2_kernel;
  const float* vb = fold ? e->b12 : w.fc2_bias;
  const void* vpk = fold ? e->pk_fold : e->pk[icap_engine::PK_FC2];
  unsigned q_launches = 0;
  const int nq_tiles = bf16 ? dense_topk_bf16_attq_tiles((int)N, c.U) : dense_topk_attq_tiles((int)N, c.U);
  auto vocab_with_attention = [&](const float* x, int lda, int rows, const LogitMasks* mk,
                                  const float* state) -> int {
    if (bf16)  // (x: fc1's fp32 output, unused - the projection reads its bf16 copy)
      return dense_topk_bf16_attq(e->fc1o16, e->pk[icap_engine::PK_FC2], w.fc2_bias, rows, c.H, c.V,
                                  c.eos_id, topk, e->rec, e->eos_x, mk, e->pf, features,
                                  w.att_v_kernel, e->att_v_bias, state, w.fc3_kernel, e->fc3_bias,
                                  e->pk[icap_engine::PK_W2], w.att_w2_bias, e->q, e->ctx, c.B, c.W,
                                  stride_g, stride_k, c.L, c.U, c.D, c.H, att_blocks, e->q_done,
                                  e->force_timeout ? ~0u : ++q_launches * (unsigned)nq_tiles,
                                  reinterpret_cast<int*>(e->q_done + 1), s, e->epf, e->pf_extreme);
    if (e->qtiles)
      return dense_topk_attq_f32(x, lda, vw, vb, rows, c.H, c.V, c.eos_id, topk, e->rec, e->eos_x, mk,
                                 e->pf, features, w.att_v_kernel, e->att_v_bias, state, w.fc3_kernel,
                                 e->fc3_bias, w.att_w2_kernel, w.att_w2_bias, e->q, e->ctx, c.B, c.W,
                                 stride_g, stride_k, c.L, c.U, c.D, c.H, att_blocks, e->q_done,
                                 e->force_timeout ? ~0u : ++q_launches * (unsigned)nq_tiles,
                                 reinterpret_cast<int*>(e->q_done + 1), s, e->epf, e->pf_extreme);
    return dense_topk_attention_f32(x, lda, vw, vb, rows, c.H, c.V, c.eos_id,
                                    topk, e->rec, e->eos_x, mk, e->pf, features, w.att_v_kernel,
                                    e->att_v_bias, state, w.fc3_kernel, e->fc3_bias,
                                    w.att_w2_kernel, w.att_w2_bias, e->ctx, c.B, c.W, stride_g,
                                    stride_k, c.L, c.U, c.D, c.H, att_blocks, s);
  };
  // bf16 mode on the side-stream schedule: the gate product [ctx | h] . W of step t+1 needs the
  // attention of step t+1 and h of step t, not the token the beam step of step t picks - only
  // the token's gate rows and the cell update do.  So the product runs on the side stream
  // behind the attention (cell_gates_bf16) and what is left on the critical path behind the
  // beam step is a pointwise kernel (cell_update_f32; the same bits as the one-launch step):
  //   main: cell_t -> fc1_t -> vocab_t -> beam_t -> cell_{t+1} ...
  //   side:     \-> query_{t+1} -> attention_{t+1} -> gates_{t+1} -/
  const bool split_cell = e->split_cell && side && bf16;
  if (e->small_loop && topk == 0 && !masks.any() && e->overlap && fold && !gru) {
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capture) != hipSuccess) (void)hipGetLastError();
    if (capture == hipStreamCaptureStatusNone) {  // (captured decodes keep the launch-per-kernel loop)
      int* err = reinterpret_cast<int*>(e->q_done + 3);
      {
        Timed tm(e, ICAP_K_VOCAB, s);
        ICAP_TRY(small_loop_decode(c, w, vw, vb, e->emb_proj, e->att_v_bias, e->fc3_bias, e->pf, e->epf,
                                   e->pf_extreme, features, e->h[0], e->h[1], e->c, e->q, e->ctx, e->logits,
                                   e->fin, e->len, e->hist_ids, e->hist_parents, e->hist_scores, e->hist_lp,
                                   e->q_done + 2, err, e->compute_units, s));
      }
      Timed tm(e, ICAP_K_BACKTRACK, s);
      e->last_err = err;
      ICAP_TRY(backtrack_guarded(e->hist_ids, e->hist_parents, e->hist_scores + (size_t)(c.T - 1) * N, e->fin,
                                 tokens, nullptr, best_score, c.T, c.B, c.W, err, s));
      return ICAP_OK;
    }
  }
  int cur = 0;
  ICAP_TRY(attention_step(s, gru ? e->h[cur] : e->c));
  if (split_cell) {
    Timed tm(e, ICAP_K_LSTM, s);
    ICAP_TRY(cell_gates_bf16(e->ctx, e->h[cur], e->pk[icap_engine::PK_LSTM], e->zgates, N, c.D, c.H, s));
  }
  for (int t = 0; t < c.T; ++t) {
    const int32_t* ids = t == 0 ? e->start_ids : e->hist_ids + (size_t)(t - 1) * N;
    if (t > 0 && side) ICAP_HIP(hipStreamWaitEvent(s, e->ev_att, 0));
    if (split_cell) {
      Timed tm(e, ICAP_K_LSTM, s);
      ICAP_TRY(cell_update_f32(gru ? 1 : 0, e->zgates, e->cell_bias, e->emb_proj, ids,
                               gru ? e->h[cur] : e->c, e->h[cur ^ 1], gru ? nullptr : e->c, N, c.H, c.V, s));
    } else {  // K6: LSTM cell on [ctx | emb(ids) | h]
      Timed tm(e, ICAP_K_LSTM, s);
      if (gru && c.precision == ICAP_PRECISION_BF16)
        ICAP_TRY(gru_step_bf16(e->ctx, e->h[cur], e->pk[icap_engine::PK_LSTM], e->cell_bias,
                               e->emb_proj, ids, e->h[cur ^ 1], N, c.D, c.H, c.V, s));
      else if (gru)
        ICAP_TRY(gru_step_f32(e->ctx, nullptr, w.embedding, ids, e->emb_proj, e->h[cur],
                              e->cell_kernel, e->cell_recurrent, e->cell_bias, e->h[cur ^ 1], N, c.D,
                              c.E, c.H, c.V, s));
      else if (c.precision == ICAP_PRECISION_BF16)
        ICAP_TRY(lstm_step_bf16(e->ctx, e->h[cur], e->c, e->pk[icap_engine::PK_LSTM], w.lstm_bias,
                                e->emb_proj, ids, e->h[cur ^ 1], e->c, N, c.D, c.H, c.V, s));
      else if (c.precision == ICAP_PRECISION_F32_SPLIT && N >= 4096)
        // (below that the exact-f32 LSTM kernel, one workgroup per CU, is as fast)
        ICAP_TRY(lstm_step_split(e->ctx, e->h[cur], e->c, e->pk[icap_engine::PK_LSTM], w.lstm_bias,
                                 e->emb_proj, ids, e->h[cur ^ 1], e->c, N, c.D, c.H, c.V, s));
      else
        ICAP_TRY(lstm_step_f32(e->ctx, nullptr, w.embedding, ids, e->emb_proj, e->h[cur], e->c,
                               w.lstm_kernel, w.lstm_recurrent_kernel, w.lstm_bias, e->h[cur ^ 1],
                               e->c, N, c.D, c.E, c.H, c.V, s));
    }
    cur ^= 1;
    if (t + 1 < c.T && side) {
      ICAP_HIP(hipEventRecord(e->ev_lstm, s));
      ICAP_HIP(hipStreamWaitEvent(e->side, e->ev_lstm, 0));
      ICAP_TRY(attention_step(e->side, gru ? e->h[cur] : e->c));
      if (split_cell) {
        Timed tm(e, ICAP_K_LSTM, e->side);
        ICAP_TRY(cell_gates_bf16(e->ctx, e->h[cur], e->pk[icap_engine::PK_LSTM], e->zgates, N, c.D,
                                 c.H, e->side));
      }
      ICAP_HIP(hipEventRecord(e->ev_att, e->side));
    }
    const float* lp_in = t == 0 ? e->lp : e->hist_lp + (size_t)(t - 1) * N;
    const int bs_time = t + lv.time_offset;  // the beam step ranks beam 0 only at time 0
    masks.step1 = bs_time + 1;
    const LogitMasks* mk = masks.any() ? &masks : nullptr;
    // wide step: outputs of the state go to the other buffer pair, the flag of this step's parity
    uint8_t* fin_o = wide ? (fin == e->fin ? e->fin2 : e->fin) : fin;
    int32_t* len_o = wide ? (len == e->len ? e->len2 : e->len) : len;
    int* flag = wide ? e->flags + (t & 1) : nullptr;
    int* flag_next = wide ? e->flags + ((t + 1) & 1) : nullptr;
    if (topk && bs_time == 0 && c.W > 1 && c.precision == ICAP_PRECISION_FP32 && !mk) {
      // Time 0 ranks the candidates of beam 0 only (beam_search.py:451-455), so fc1 and the
      // vocab projection run on every W-th row (row stride W*H) and write one record row per
      // image; the recurrent state of all rows was still advanced above.
      if (!fold) {
        Timed tm(e, ICAP_K_FC1, s);
        ICAP_TRY(dense_f32_strided(e->h[cur], c.W * c.H, w.fc1_kernel, w.fc1_bias, e->fc1o, c.B, c.H,
                                   c.H, ICAP_ACT_NONE, s));
      }
      const float* x0 = fold ? e->h[cur] : e->fc1o;  // beam 0's rows
      const int lda0 = fold ? c.W * c.H : c.H;
      if (beam_in_launch) {
        Timed tm(e, ICAP_K_VOCAB, s);
        ICAP_TRY(dense_topk_beam_f32(x0, lda0, vw, vb, c.B, c.H, c.V, c.eos_id,
                                     topk, e->rec, e->eos_x, lp_in, fin, len, 0,
                                     c.length_penalty_weight, e->hist_scores, e->hist_ids,
                                     e->hist_parents, e->hist_lp, fin, len, c.B, c.W, 1, done0, 1,
                                     beam_err, s));
        if (t + 1 < c.T && !e->overlap) ICAP_TRY(attention_step(s, gru ? e->h[cur] : e->c));
        continue;
      }
      {
        Timed tm(e, ICAP_K_VOCAB, s);
        if (merged && t + 1 < c.T)
          ICAP_TRY(vocab_with_attention(x0, lda0, c.B, nullptr, gru ? e->h[cur] : e->c));
        else
          ICAP_TRY(dense_topk_f32_strided(x0, lda0, vw, vb, c.B, c.H, c.V,
                                          c.eos_id, topk, e->rec, e->eos_x, nullptr, s));
      }
      if (t + 1 < c.T && !e->overlap) ICAP_TRY(attention_step(s, gru ? e->h[cur] : e->c));
      Timed tm(e, ICAP_K_BEAM, s);
      if (wide) {
        ICAP_TRY(beam_step_from_records_wide_f32(
            e->rec, e->eos_x, dense_topk_half_tiles(c.V), topk, lp_in, fin, len, 0, c.eos_id,
            c.length_penalty_weight, e->hist_scores, e->hist_ids, e->hist_parents, e->hist_lp, fin_o,
            len_o, c.B, c.W, c.V, flag, s, 1));
        if (e->wide_rescues_forced) {
          hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, s, flag, 1, 1);
          ICAP_LAUNCH_CHECK("fill_i32_kernel");
        }
        // behind it, switched off on the device unless the flag is up: the same step on the
        // full logits of beam 0 ([B,1,V]; the state before step 0 is all zeros, so its first
        // B entries serve as the [B,1] state)
        ICAP_TRY(dense_f32_flagged(x0, lda0, vw, vb, e->logits, c.B, c.H, c.V, flag, s));
        ICAP_TRY(beam_step_rows_f32(e->logits, lp_in, fin, len, 0, 0, c.eos_id,
                                    c.length_penalty_weight, 0, e->hist_scores, e->hist_ids,
                                    e->hist_parents, e->hist_lp, fin_o, len_o, c.B, 1, c.W, c.V,
                                    e->beam_ws, e->beam_ws_bytes, s, flag, flag_next));
        fin = fin_o;
        len = len_o;
        continue;
      }
      ICAP_TRY(beam_step_from_records_f32(
          e->rec, e->eos_x, dense_topk_half_tiles(c.V), topk, lp_in, fin, len, 0, c.eos_id,
          c.length_penalty_weight, e->hist_scores, e->hist_ids, e->hist_parents, e->hist_lp, fin,
          len, c.B, c.W, c.V, s, 1));
      continue;
    }
    // bf16 mode, fused projection: fc1 also writes its output as bf16 and the projection reads
    // that (the values the projection's own staging conversion would produce)
    const bool a16 = topk != 0 && c.precision == ICAP_PRECISION_BF16 && e->a16;
    const int a16_blocked = a16 && dense_topk_bf16_a16_blocked(N, c.H, c.V, topk) ? 1 : 0;
    const float* vx = fold ? e->h[cur] : e->fc1o;  // [N,H]
    if (!fold) {  // K7: fc1
      Timed tm(e, ICAP_K_FC1, s);
      if (a16)
        ICAP_TRY(dense_bf16_dual(e->h[cur], e->pk[icap_engine::PK_FC1], w.fc1_bias, e->fc1o, e->fc1o16,
                                 a16_blocked, N, c.H, c.H, ICAP_ACT_NONE, s));
      else
        ICAP_TRY(engine_dense(e, icap_engine::PK_FC1, e->h[cur], w.fc1_kernel, w.fc1_bias, e->fc1o, N,
                              c.H, c.H, ICAP_ACT_NONE, s));
    }
    if (topk) {
      // K8+K9 fused: the projection keeps per-row top-k records, the logits never reach HBM
      if (t + 1 < c.T && side && vocab_gate && !((e->timing >> ICAP_K_VOCAB) & 1u))
        ICAP_HIP(hipEventRecord(e->ev_gate, s));
      if (beam_in_launch && c.precision == ICAP_PRECISION_FP32) {
        Timed tm(e, ICAP_K_VOCAB, s);
        ICAP_TRY(dense_topk_beam_f32(vx, c.H, vw, vb, N, c.H, c.V, c.eos_id,
                                     topk, e->rec, e->eos_x, lp_in, fin, len, bs_time,
                                     c.length_penalty_weight, e->hist_scores + (size_t)t * N,
                                     e->hist_ids + (size_t)t * N, e->hist_parents + (size_t)t * N,
                                     e->hist_lp + (size_t)t * N, fin, len, c.B, c.W, 0, done1,
                                     ++launches1, beam_err, s));
        if (t + 1 < c.T && !e->overlap) ICAP_TRY(attention_step(s, gru ? e->h[cur] : e->c));
        continue;
      }
      {
        Timed tm(e, ICAP_K_VOCAB, s);
        if (bf16 && merged && t + 1 < c.T)
          ICAP_TRY(vocab_with_attention(vx, c.H, N, mk, gru ? e->h[cur] : e->c));
        else if (a16)
          ICAP_TRY(dense_topk_bf16_a16(e->fc1o16, a16_blocked, e->pk[icap_engine::PK_FC2], w.fc2_bias,
                                       N, c.H, c.V, c.eos_id, topk, e->rec, e->eos_x, mk, s));
        else if (c.precision == ICAP_PRECISION_BF16)
          ICAP_TRY(dense_topk_bf16(e->fc1o, e->pk[icap_engine::PK_FC2], w.fc2_bias, N, c.H, c.V,
                                   c.eos_id, topk, e->rec, e->eos_x, mk, s));
        else if (c.precision == ICAP_PRECISION_F32_SPLIT)
          ICAP_TRY(dense_topk_split(vx, vpk, vb, N, c.H, c.V,
                                    c.eos_id, topk, e->rec, e->eos_x, mk, s));
        else if (merged && t + 1 < c.T)
          ICAP_TRY(vocab_with_attention(vx, c.H, N, mk, gru ? e->h[cur] : e->c));
        else
          ICAP_TRY(dense_topk_f32(vx, vw, vb, N, c.H, c.V, c.eos_id,
                                  topk, e->rec, e->eos_x, mk, s));
      }
      if (t + 1 < c.T && !e->overlap) ICAP_TRY(attention_step(s, gru ? e->h[cur] : e->c));
      Timed tm(e, ICAP_K_BEAM, s);
      if (wide) {
        ICAP_TRY(beam_step_from_records_wide_f32(
            e->rec, e->eos_x, dense_topk_half_tiles(c.V), topk, lp_in, fin, len, bs_time, c.eos_id,
            c.length_penalty_weight, e->hist_scores + (size_t)t * N, e->hist_ids + (size_t)t * N,
            e->hist_parents + (size_t)t * N, e->hist_lp + (size_t)t * N, fin_o, len_o, c.B, c.W,
            c.V, flag, s));
        if (e->wide_rescues_forced) {
          hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, s, flag, 1, 1);
          ICAP_LAUNCH_CHECK("fill_i32_kernel");
        }
        // behind it, switched off on the device unless the flag is up: the unfused step
        ICAP_TRY(dense_f32_flagged(vx, c.H, vw, vb, e->logits, N, c.H, c.V, flag, s));
        ICAP_TRY(beam_step_rows_f32(e->logits, lp_in, fin, len, bs_time, 0, c.eos_id,
                                    c.length_penalty_weight, 0, e->hist_scores + (size_t)t * N,
                                    e->hist_ids + (size_t)t * N, e->hist_parents + (size_t)t * N,
                                    e->hist_lp + (size_t)t * N, fin_o, len_o, c.B, c.W, c.W, c.V,
                                    e->beam_ws, e->beam_ws_bytes, s, flag, flag_next));
        fin = fin_o;
        len = len_o;
        continue;
      }
      ICAP_TRY(beam_step_from_records_f32(
          e->rec, e->eos_x, dense_topk_half_tiles(c.V), topk, lp_in, fin, len, bs_time,
          c.eos_id,
          c.length_penalty_weight, e->hist_scores + (size_t)t * N, e->hist_ids + (size_t)t * N,
          e->hist_parents + (size_t)t * N, e->hist_lp + (size_t)t * N, fin, len, c.B, c.W,
          c.V, s));
      continue;
    }
    {  // K8: vocab projection
      Timed tm(e, ICAP_K_VOCAB, s);
      ICAP_TRY(engine_dense(e, icap_engine::PK_FC2, vx, vw, vb, e->logits, N, c.H, c.V, ICAP_ACT_NONE,
                            s, vpk));
      if (mk) ICAP_TRY(logit_masks_f32(e->logits, N, c.V, masks, s));
    }
    if (t + 1 < c.T && !e->overlap) ICAP_TRY(attention_step(s, gru ? e->h[cur] : e->c));
    {  // K9: beam step.  Outputs and the accumulated log-probs are recorded per step; the
       // log-probs of step t-1 are the state input of step t (zeros at t = 0).
      Timed tm(e, ICAP_K_BEAM, s);
      ICAP_TRY(beam_step_f32(e->logits, lp_in, fin, len, bs_time, 0, c.eos_id,
                             c.length_penalty_weight, 0, e->hist_scores + (size_t)t * N,
                             e->hist_ids + (size_t)t * N, e->hist_parents + (size_t)t * N,
                             e->hist_lp + (size_t)t * N, fin, len, c.B, c.W, c.V,
                             e->beam_ws, e->beam_ws_bytes, s));
    }
  }
  {  // K10
    Timed tm(e, ICAP_K_BACKTRACK, s);
    // (a hand-over inside a projection launch whose wait ran out invalidates the result: every
    // token -1, every score NaN, and icap_engine_decode_status reports it)
    e->last_err = beam_in_launch ? beam_err
                  : (merged && e->qtiles) ? reinterpret_cast<int*>(e->q_done + 1) : nullptr;
    ICAP_TRY(backtrack_guarded(e->hist_ids, e->hist_parents, e->hist_scores + (size_t)(c.T - 1) * N,
                               fin, tokens, nullptr, best_score, c.T, c.B, c.W, e->last_err, s));
  }
  return ICAP_OK;
}

}  // namespace
}  // namespace icap

using namespace icap;

extern "C" {

int icap_version(void) { return 110; }

size_t icap_last_error(char* buf, size_t len) {
  const std::string& m = last_error();
  if (buf && len) {
    const size_t n = m.size() < len - 1 ? m.size() : len - 1;
    memcpy(buf, m.data(), n);
    buf[n] = 0;
  }
  return m.size();
}

int icap_dense_f32(const float* x, const float* w, const float* bias, float* y, int M, int K,
                   int N, int act, icap_stream_t stream) {
  return dense_f32(x, w, bias, y, M, K, N, act, (hipStream_t)stream);
}

size_t icap_packed_weight_bytes(int K, int N) { return packed_weight_bytes(K, N); }

int icap_pack_weight_bf16(const float* w, void* packed, int K_src, int N, int k_offset, int K_total,
                          int gate_units, icap_stream_t stream) {
  return pack_weight_bf16(w, packed, K_src, N, k_offset, K_total, gate_units, (hipStream_t)stream);
}

int icap_dense_bf16(const float* x, const void* packed_w, const float* bias, float* y, int M, int K,
                    int N, int act, icap_stream_t stream) {
  return dense_bf16(x, packed_w, bias, y, M, K, N, act, (hipStream_t)stream);
}

int icap_lstm_step_bf16(const float* ctx, const float* h, const float* c, const void* packed_w,
                        const float* bias, const float* emb_proj, const int32_t* ids, float* h_out,
                        float* c_out, int N, int C, int H, int V, icap_stream_t stream) {
  return lstm_step_bf16(ctx, h, c, packed_w, bias, emb_proj, ids, h_out, c_out, N, C, H, V,
                        (hipStream_t)stream);
}

size_t icap_packed_split_bytes(int K, int N) { return packed_split_bytes(K, N); }

int icap_pack_weight_split(const float* w, void* packed, int K_src, int N, int k_offset,
                           int K_total, int gate_units, icap_stream_t stream) {
  return pack_weight_split(w, packed, K_src, N, k_offset, K_total, gate_units, (hipStream_t)stream);
}

int icap_dense_split(const float* x, const void* packed_w, const float* bias, float* y, int M,
                     int K, int N, int act, icap_stream_t stream) {
  return dense_split(x, packed_w, bias, y, M, K, N, act, (hipStream_t)stream);
}

int icap_dense_topk_split(const float* x, const void* packed_w, const float* bias, int M, int K,
                          int N, int eos_token, int top_k, float* records, float* eos_logit,
                          icap_stream_t stream) {
  return dense_topk_split(x, packed_w, bias, M, K, N, eos_token, top_k, records, eos_logit, nullptr,
                          (hipStream_t)stream);
}

int icap_lstm_step_split(const float* ctx, const float* h, const float* c, const void* packed_w,
                         const float* bias, const float* emb_proj, const int32_t* ids,
                         float* h_out, float* c_out, int N, int C, int H, int V,
                         icap_stream_t stream) {
  return lstm_step_split(ctx, h, c, packed_w, bias, emb_proj, ids, h_out, c_out, N, C, H, V,
                         (hipStream_t)stream);
}

int icap_pool_features_f32(const float* features, const float* mask, float* pooled, int G, int L,
                           int D, icap_stream_t stream) {
  return pool_features_f32(features, mask, pooled, G, L, D, (hipStream_t)stream);
}

int icap_expand_rows_f32(const float* src, float* dst, int B, int W, int H, float scale,
                         int tile_mode, icap_stream_t stream) {
  return expand_rows_f32(src, dst, B, W, H, scale, tile_mode, (hipStream_t)stream);
}

int icap_embedding_f32(const float* table, const int32_t* ids, float* out, int N, int V, int E,
                       icap_stream_t stream) {
  return embedding_f32(table, ids, out, N, V, E, (hipStream_t)stream);
}

int icap_attention_f32(const float* pf, const float* features, const float* q, const float* v,
                       float bv, const float* score_bias, const float* gate_state,
                       const float* gate_w, float gate_b, float* ctx, float* weights, int G,
                       int rows_per_group, int row_stride_g, int row_stride_k, int L, int U, int D,
                       int H, icap_stream_t stream) {
  return attention_f32(pf, features, q, v, bv, score_bias, gate_state, gate_w, gate_b, ctx, weights,
                       G, rows_per_group, row_stride_g, row_stride_k, L, U, D, H, 0,
                       (hipStream_t)stream);
}

int icap_attention_act_f32(const float* pf, const float* features, const float* q, const float* v,
                           float bv, const float* score_bias, const float* gate_state,
                           const float* gate_w, float gate_b, float* ctx, float* weights, int G,
                           int rows_per_group, int row_stride_g, int row_stride_k, int L, int U,
                           int D, int H, int activation, icap_stream_t stream) {
  return attention_act_f32(pf, features, q, v, bv, score_bias, gate_state, gate_w, gate_b, ctx,
                           weights, G, rows_per_group, row_stride_g, row_stride_k, L, U, D, H, 0,
                           activation, (hipStream_t)stream);
}

int icap_lstm_step_f32(const float* ctx, const float* emb, const float* table, const int32_t* ids,
                       const float* emb_proj, const float* h, const float* c, const float* kernel,
                       const float* recurrent_kernel, const float* bias, float* h_out,
                       float* c_out, int N, int C, int E, int H, int V, icap_stream_t stream) {
  return lstm_step_f32(ctx, emb, table, ids, emb_proj, h, c, kernel, recurrent_kernel, bias, h_out,
                       c_out, N, C, E, H, V, (hipStream_t)stream);
}

int icap_gru_pack_f32(const float* kernel, const float* recurrent_kernel, const float* bias,
                      float* kernel4, float* recurrent4, float* bias4, int CE, int H,
                      icap_stream_t stream) {
  return gru_pack_f32(kernel, recurrent_kernel, bias, kernel4, recurrent4, bias4, CE, H,
                      (hipStream_t)stream);
}

int icap_gru_step_bf16(const float* ctx, const float* h, const void* packed_w, const float* bias4,
                       const float* emb_proj, const int32_t* ids, float* h_out, int N, int C, int H,
                       int V, icap_stream_t stream) {
  return gru_step_bf16(ctx, h, packed_w, bias4, emb_proj, ids, h_out, N, C, H, V, (hipStream_t)stream);
}

int icap_gru_step_f32(const float* ctx, const float* emb, const float* table, const int32_t* ids,
                      const float* emb_proj, const float* h, const float* kernel4,
                      const float* recurrent4, const float* bias4, float* h_out, int N, int C, int E,
                      int H, int V, icap_stream_t stream) {
  return gru_step_f32(ctx, emb, table, ids, emb_proj, h, kernel4, recurrent4, bias4, h_out, N, C, E,
                      H, V, (hipStream_t)stream);
}

int icap_beam_step_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                       const int32_t* lengths, int time, int start_offset, int eos_token,
                       double length_penalty_weight, int input_is_log_prob, float* scores,
                       int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                       uint8_t* next_finished, int32_t* next_lengths, int B, int W, int V,
                       void* workspace, size_t workspace_bytes, icap_stream_t stream) {
  return beam_step_f32(logits, log_probs, finished, lengths, time, start_offset, eos_token,
                       length_penalty_weight, input_is_log_prob, scores, predicted_ids, parent_ids,
                       next_log_probs, next_finished, next_lengths, B, W, V, workspace,
                       workspace_bytes, (hipStream_t)stream);
}

int icap_beam_step_rows_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                            const int32_t* lengths, int time, int start_offset, int eos_token,
                            double length_penalty_weight, int input_is_log_prob, float* scores,
                            int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                            uint8_t* next_finished, int32_t* next_lengths, int B, int R, int W,
                            int V, void* workspace, size_t workspace_bytes, icap_stream_t stream) {
  return beam_step_rows_f32(logits, log_probs, finished, lengths, time, start_offset, eos_token,
                            length_penalty_weight, input_is_log_prob, scores, predicted_ids,
                            parent_ids, next_log_probs, next_finished, next_lengths, B, R, W, V,
                            workspace, workspace_bytes, (hipStream_t)stream);
}

size_t icap_beam_step_workspace_bytes(int B, int W) { return beam_step_workspace_bytes(B, W); }

int icap_beam_step_static_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                              const int32_t* lengths, int time, int start_offset, int eos_token,
                              double length_penalty_weight, int input_is_log_prob, float* scores,
                              int32_t* predicted_ids, int32_t* parent_ids, float* next_log_probs,
                              uint8_t* next_finished, int32_t* next_lengths, int B, int W, int V,
                              icap_stream_t stream) {
  return beam_step_static_f32(logits, log_probs, finished, lengths, time, start_offset, eos_token,
                              length_penalty_weight, input_is_log_prob, scores, predicted_ids,
                              parent_ids, next_log_probs, next_finished, next_lengths, B, W, V,
                              (hipStream_t)stream);
}

int icap_sample_categorical_f32(const float* logits, const float* u, const int32_t* row_len,
                                int step, int start_token, int eos_token, int pad_token,
                                int32_t* samples, int N, int V, icap_stream_t stream) {
  return sample_categorical_f32(logits, u, row_len, step, start_token, eos_token, pad_token,
                                samples, N, V, (hipStream_t)stream);
}

int icap_beam_scores_f32(const float* logits, const float* log_probs, const uint8_t* finished,
                         const int32_t* lengths, int eos_token, double length_penalty_weight,
                         int input_is_log_prob, float* scores, float* totals, int B, int W, int V,
                         icap_stream_t stream) {
  return beam_scores_f32(logits, log_probs, finished, lengths, eos_token, length_penalty_weight,
                         input_is_log_prob, scores, totals, B, W, V, (hipStream_t)stream);
}

int icap_beam_gather_f32(const float* totals, const uint8_t* finished, const int32_t* lengths,
                         const int32_t* word_indices, int eos_token, int32_t* predicted_ids,
                         int32_t* parent_ids, float* next_log_probs, uint8_t* next_finished,
                         int32_t* next_lengths, int B, int W, int V, int K, icap_stream_t stream) {
  return beam_gather_f32(totals, finished, lengths, word_indices, eos_token, predicted_ids,
                         parent_ids, next_log_probs, next_finished, next_lengths, B, W, V, K,
                         (hipStream_t)stream);
}

int icap_mask_probs_f32(const float* probs, const uint8_t* finished, float* out, int eos_token,
                        int N, int V, icap_stream_t stream) {
  return mask_probs_f32(probs, finished, out, eos_token, N, V, (hipStream_t)stream);
}

int icap_hyp_score_f32(const float* log_probs, const int32_t* lengths, double length_penalty_weight,
                       float* out, long long n, icap_stream_t stream) {
  return hyp_score_f32(log_probs, lengths, length_penalty_weight, out, n, (hipStream_t)stream);
}

int icap_batch_gather_f32(const float* a, const int32_t* idx, float* out, int B, int X, int Y,
                          icap_stream_t stream) {
  return batch_gather_f32(a, idx, out, B, X, Y, (hipStream_t)stream);
}

int icap_dense_topk_half_tiles(int N) { return dense_topk_half_tiles(N); }
int icap_dense_topk_record_floats(int top_k) { return dense_topk_record_floats(top_k); }

int icap_dense_topk_f32(const float* x, const float* w, const float* bias, int M, int K, int N,
                        int eos_token, int top_k, float* records, float* eos_logit,
                        icap_stream_t stream) {
  return dense_topk_f32(x, w, bias, M, K, N, eos_token, top_k, records, eos_logit, nullptr,
                        (hipStream_t)stream);
}

int icap_beam_step_from_records_f32(const float* records, const float* eos_logit, int half_tiles,
                                    int top_k, const float* log_probs, const uint8_t* finished,
                                    const int32_t* lengths, int time, int eos_token,
                                    double length_penalty_weight, float* scores,
                                    int32_t* predicted_ids, int32_t* parent_ids,
                                    float* next_log_probs, uint8_t* next_finished,
                                    int32_t* next_lengths, int B, int W, int V,
                                    icap_stream_t stream) {
  return beam_step_from_records_f32(records, eos_logit, half_tiles, top_k, log_probs, finished,
                                    lengths, time, eos_token, length_penalty_weight, scores,
                                    predicted_ids, parent_ids, next_log_probs, next_finished,
                                    next_lengths, B, W, V, (hipStream_t)stream);
}

int icap_beam_step_from_records_wide_f32(const float* records, const float* eos_logit,
                                         int half_tiles, int top_k, const float* log_probs,
                                         const uint8_t* finished, const int32_t* lengths, int time,
                                         int eos_token, double length_penalty_weight, float* scores,
                                         int32_t* predicted_ids, int32_t* parent_ids,
                                         float* next_log_probs, uint8_t* next_finished,
                                         int32_t* next_lengths, int B, int W, int V,
                                         int32_t* overflow, int compact_first,
                                         icap_stream_t stream) {
  return beam_step_from_records_wide_f32(records, eos_logit, half_tiles, top_k, log_probs, finished,
                                         lengths, time, eos_token, length_penalty_weight, scores,
                                         predicted_ids, parent_ids, next_log_probs, next_finished,
                                         next_lengths, B, W, V, overflow, (hipStream_t)stream,
                                         compact_first);
}

int icap_backtrack(const int32_t* ids, const int32_t* parents, const float* last_scores,
                   const uint8_t* last_finished, int32_t* tokens, int32_t* best_beam,
                   float* best_score, int T, int B, int W, icap_stream_t stream) {
  return backtrack(ids, parents, last_scores, last_finished, tokens, best_beam, best_score, T, B, W,
                   (hipStream_t)stream);
}

int icap_engine_create(const icap_config* config, const icap_weights* weights, icap_engine** out) {
  ICAP_REQUIRE(config && weights && out, "icap_engine_create: null argument");
  *out = nullptr;
  ICAP_TRY(check_config(*config));
  const void* const* wp = reinterpret_cast<const void* const*>(weights);
  for (size_t i = 0; i < sizeof(icap_weights) / sizeof(void*); ++i) {
    const bool optional = config->cell == ICAP_CELL_GRU &&
                          (&wp[i] == reinterpret_cast<const void* const*>(&weights->init_c_kernel) ||
                           &wp[i] == reinterpret_cast<const void* const*>(&weights->init_c_bias));
    ICAP_REQUIRE(optional || wp[i] != nullptr, "icap_engine_create: weight pointer %zu is null", i);
  }
  icap_engine* e = new (std::nothrow) icap_engine();
  ICAP_REQUIRE(e, "icap_engine_create: out of host memory");
  e->cfg = *config;
  e->w = *weights;
  const icap_config& c = e->cfg;
  const size_t N = (size_t)c.B * c.W, BL = (size_t)c.B * c.L, TN = (size_t)c.T * N;
  Carver cv;
  const size_t o_features = cv.take<float>(BL * c.D), o_pf = cv.take<float>(BL * c.U),
               o_pooled = cv.take<float>((size_t)c.B * c.D),
               o_init_h = cv.take<float>((size_t)c.B * c.H),
               o_init_c = cv.take<float>((size_t)c.B * c.H), o_h0 = cv.take<float>(N * c.H),
               o_h1 = cv.take<float>(N * c.H), o_c = cv.take<float>(N * c.H),
               o_q = cv.take<float>(N * c.U), o_ctx = cv.take<float>(N * c.D),
               o_fc1 = cv.take<float>(N * c.H), o_logits = cv.take<float>(N * c.V),  // (unused by the fused top-k path)
               o_lp = cv.take<float>(N), o_hs = cv.take<float>(TN), o_hl = cv.take<float>(TN),
               o_len = cv.take<int32_t>(N), o_hi = cv.take<int32_t>(TN),
               o_hp = cv.take<int32_t>(TN), o_si = cv.take<int32_t>(N), o_fin = cv.take<uint8_t>(N),
               o_bws = cv.take<char>(beam_step_workspace_bytes(c.B, c.W)),
               o_ep = cv.take<float>((size_t)c.V * 4 * c.H);
  const size_t o_rl = cv.take<int32_t>(N);
  const bool fold_able = c.precision == ICAP_PRECISION_FP32 || c.precision == ICAP_PRECISION_F32_SPLIT;
  const size_t o_w12 = cv.take<float>(fold_able ? (size_t)c.H * c.V : 0),
               o_b12 = cv.take<float>(fold_able ? (size_t)c.V : 0);
  const size_t o_fin2 = cv.take<uint8_t>(N), o_len2 = cv.take<int32_t>(N), o_flags = cv.take<char>(256);
  e->tile_rows = ceil_div((int)N, 128);
  const size_t o_trd = cv.take<unsigned>(2 * e->tile_rows + 1);
  const size_t o_qd = cv.take<unsigned>(64);
  // (ICAP_ATT_TABLE=0, read per engine: no table, the attention kernels evaluate tanh directly)
  const char* table_env = getenv("ICAP_ATT_TABLE");
  const bool att_table = !(table_env && table_env[0] == '0') && (BL * c.U) % 4 == 0;
  // bf16 mode: the table and the features as bf16 for the side-stream attention kernel; the
  // float32 table only for the bf16 form of the merged launch (ICAP_MERGED=1), which reads that
  const char* a16_env0 = getenv("ICAP_ATT_BF16");
  const char* merged_env0 = getenv("ICAP_MERGED");
  const bool att16 = att_table && c.precision == ICAP_PRECISION_BF16 && a16_env0 && a16_env0[0] == '1' &&
                     !(merged_env0 && merged_env0[0] == '1') && (BL * c.D) % 4 == 0 && c.U % 4 == 0 &&
                     c.D % 4 == 0;
  const size_t o_epf = cv.take<float>(att_table && !att16 ? BL * c.U : 0), o_ext = cv.take<int>(64);
  const size_t o_epf16 = cv.take<uint16_t>(att16 ? BL * c.U : 0), o_f16 = cv.take<uint16_t>(att16 ? BL * c.D : 0);
  const size_t o_fc116 =  // (rows rounded up to the 32-row operand blocks of the resident kernel)
      cv.take<uint16_t>(c.precision == ICAP_PRECISION_BF16 ? (N + 31) / 32 * 32 * c.H : 0);
  // (ICAP_BF16_SPLIT_CELL=1, read per engine: the cell step of the bf16 side-stream schedule as
  // gate product + cell update, see decode_impl.  Opt-in: measured 179k against 189k captions/s
  // at B = 512, beam 5 and 221k against 233k at beam 3 - the gate product lengthens the side
  // chain (query, attention, product) to the length of the main one and competes with the
  // projection's tail and the beam step for the L2.)
  const char* sc_env = getenv("ICAP_BF16_SPLIT_CELL");
  const bool split_cell = c.precision == ICAP_PRECISION_BF16 && sc_env && sc_env[0] == '1';
  const size_t o_zg = cv.take<float>(split_cell ? N * 4 * c.H : 0);
  const bool gru = c.cell == ICAP_CELL_GRU;
  const size_t o_gk = cv.take<float>(gru ? (size_t)(c.D + c.E) * 4 * c.H : 0),
               o_gr = cv.take<float>(gru ? (size_t)c.H * 4 * c.H : 0),
               o_gb = cv.take<float>(gru ? (size_t)4 * c.H : 0);
  // packed bf16 weights: {K, N} per matrix
  const int pk_dims[icap_engine::PK_COUNT][2] = {
      {c.F, c.D}, {c.D, c.U}, {c.D, c.H}, {c.D, c.H}, {c.H, c.U},
      {c.D + c.H, 4 * c.H}, {c.E, 4 * c.H}, {c.H, c.H}, {c.H, c.V}};
  // Fused vocab projection + beam step when the shape allows it (tuning knob
  // ICAP_FUSED_TOPK=0 keeps the logits buffer and the separate beam kernels).
  {
    static const char* fused_env = getenv("ICAP_FUSED_TOPK");
    const int tk = c.W <= 1 ? 1 : (c.W <= 3 ? 3 : 5);
    const bool gemm_ok =
        c.precision == ICAP_PRECISION_BF16
            ? dense_topk_bf16_supported(weights->fc2_kernel, (int)N, c.H, c.V, tk)
        : c.precision == ICAP_PRECISION_F32_SPLIT
            ? dense_topk_split_supported(weights->fc2_kernel, (int)N, c.H, c.V, tk)
            : dense_topk_supported(weights->fc2_kernel, weights->fc2_kernel, (int)N, c.H, c.V, tk);
    const bool shape_ok =
        c.W <= 5 && gemm_ok && beam_records_supported(c.W, dense_topk_half_tiles(c.V), tk);
    e->topk_able = shape_ok ? tk : 0;
    // wider beams (exact-f32 only): records of 5 with the proof-or-repeat beam step
    if (c.W > 5 && c.precision == ICAP_PRECISION_FP32 && gemm_ok &&
        dense_f32_flagged_supported(weights->fc2_kernel, weights->fc2_kernel, c.H, (int)N, c.H, c.V) &&
        (long long)c.W * dense_topk_half_tiles(c.V) < (1LL << 24)) {
      e->wide = true;
      e->topk_able = 5;
      const char* force_env = getenv("ICAP_FORCE_WIDE_RESCUE");  // test hook, read per engine
      e->wide_rescues_forced = force_env && force_env[0] == '1';
    }
    // The fused projection works in 128x128 tiles; below about three of them per compute
    // unit the plain projection (64x64 tiles, or the small-batch form) plus the unfused beam
    // step is faster: measured at beam 3, B = 32 / 128 / 256 / 384 / 512: 15.5k / 42k / 65k /
    // 80k / 96k captions/s fused against 18k / 50k / 73k / 79k / 93k unfused.
    const long tiles128 = (long)ceil_div((int)N, 128) * ceil_div(c.V, 128);
    const bool fused_default = tiles128 >= 700;
    e->topk = fused_env ? (fused_env[0] == '0' ? 0 : e->topk_able)
                        : (fused_default ? e->topk_able : 0);
  }
  const size_t o_rec = cv.take<float>(e->topk_able ? N * dense_topk_half_tiles(c.V) *
                                                         dense_topk_record_floats(e->topk_able)
                                                   : 0),
               o_eosx = cv.take<float>(N);
  size_t o_pk[icap_engine::PK_COUNT] = {};
  if (c.precision == ICAP_PRECISION_BF16)
    for (int i = 0; i < icap_engine::PK_COUNT; ++i)
      o_pk[i] = cv.take<char>(packed_weight_bytes(pk_dims[i][0], pk_dims[i][1]));
  if (c.precision == ICAP_PRECISION_F32_SPLIT)
    for (int i = 0; i < icap_engine::PK_COUNT; ++i)
      o_pk[i] = cv.take<char>(packed_split_bytes(pk_dims[i][0], pk_dims[i][1]));
  const size_t o_pkf =
      cv.take<char>(c.precision == ICAP_PRECISION_F32_SPLIT ? packed_split_bytes(c.H, c.V) : 0);
  e->ws_bytes = (cv.off + 255) / 256 * 256;
  hipError_t err = hipGetDevice(&e->device);
  if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->ws), e->ws_bytes);
  if (err == hipSuccess) err = shared_side_stream(e->device, &e->side);  // (owned by the library, not the engine)
  if (err == hipSuccess) {
    // beside the exact-f32 vocab projection the attention kernel gets one workgroup per
    // compute unit, which leaves the projection its two workgroups per CU (96.8k captions/s
    // against 92.7k uncapped at the headline shape).  The bf16 projection is short enough
    // that a capped attention becomes the critical path (114k capped, 129k uncapped at
    // B=512, W=5), so it runs uncapped there.  Tuning knob ICAP_SIDE_ATT_WGS: 0 = uncapped.
    hipDeviceProp_t prop;
    err = hipGetDeviceProperties(&prop, e->device);
    if (err == hipSuccess) e->compute_units = prop.multiProcessorCount;
    static const char* cap_env = getenv("ICAP_SIDE_ATT_WGS");
    if (err == hipSuccess)
      // (bf16 mode: uncapped, except beams of 4 and 5 whose images divide evenly over one
      // workgroup per compute unit - B = 512, beam 5: 188k against 181k captions/s, beam 4: 215k
      // against 206k; B = 384, beam 5: 151k against 157k; beam 3: 202k against 225k)
      e->side_att_wgs = cap_env ? atoi(cap_env)
                        : c.precision != ICAP_PRECISION_BF16 ? prop.multiProcessorCount
                        : (c.W >= 4 && c.B % prop.multiProcessorCount == 0) ? prop.multiProcessorCount
                                                                             : 0;
  }
  {
    // The next step's query + attention inside the projection's launch.  With the folded
    // output kernel nothing stands between the cell kernel and the projection any more, so on
    // the side stream the attention (behind its query kernel) would always arrive after the
    // projection has taken every slot; in the projection's own launch its workgroups come
    // first by construction and the query kernel, two event dependencies and the launch race
    // are gone: 107.8k against 100.8k captions/s at the headline shape (with fc1 as a launch
    // of its own the two schedules are equal, 101.6k against 101.4k).  ICAP_MERGED=0|1
    // overrides (read when the engine is created).
    const char* merged_env = getenv("ICAP_MERGED");
    const char* fold_env0 = getenv("ICAP_FOLD_FC");
    const bool fold_on = fold_able && (fold_env0 ? fold_env0[0] != '0' : c.precision == ICAP_PRECISION_FP32);
    const bool want = merged_env ? merged_env[0] == '1' : fold_on;
    const bool base_ok = want && c.precision == ICAP_PRECISION_FP32 && e->topk_able != 0 &&
                         aligned16(weights->att_v_kernel) && aligned16(weights->att_w2_kernel);
    // the query projection as tiles of the same launch (ICAP_MERGED_QTILES=0: inside every
    // attention workgroup instead, beams up to 5); wide beams take the tiles form only
    const char* qt_env = getenv("ICAP_MERGED_QTILES");
    const char* ft_env = getenv("ICAP_MERGED_FORCE_TIMEOUT");  // test hook, read per engine
    e->force_timeout = ft_env && ft_env[0] == '1';
    e->qtiles = base_ok && !(qt_env && qt_env[0] == '0') &&
                dense_topk_attq_supported(c.W, c.L, c.U, c.D, c.H, e->topk_able);
    e->merged = e->qtiles ||
                (base_ok && !e->wide &&
                 dense_topk_attention_supported(c.W, c.L, c.U, c.D, c.H, e->topk_able));
    // With the merged launch the fused path wins from about a hundred beam rows upwards (beam 3:
    // B = 32 25.2k against 20.6k captions/s, B = 64 45k against 37k, B = 256 83k against 77k;
    // greedy: B = 128 103k against 88k, B = 1024 244k against 215k; B*W <= 64 stays 4-8 %
    // faster on the plain projection + unfused step), not only from three tiles per compute
    // unit as on the side-stream schedule.  The bf16 and split modes, on the side stream, gain
    // from the same size on (bf16 beam 3: B = 128 90k against 83k, B = 256 151k against 127k,
    // greedy B = 1024 478k against 379k; split beam 3: B = 128 64k against 55k, B = 384 101k
    // against 89k; equal within 1 % at 80-128 rows, the unfused step ahead below).
    {
      static const char* fused_env2 = getenv("ICAP_FUSED_TOPK");
      if (!fused_env2 && N > 64 && !e->wide && (e->merged || c.precision != ICAP_PRECISION_FP32))
        e->topk = e->topk_able;
      // wide beams with the merged launch: from 512 rows (beam 16: B = 32 11.1k against 9.9k,
      // B = 64 16.0k against 13.9k; beam 10: B = 64 19.6k against 18.6k, B = 32 12.7k against 13.8k)
      if (!fused_env2 && e->wide && e->merged && N >= 512) e->topk = e->topk_able;
    }
    // bf16 mode: the same arrangement around the bf16 record projection (query tiles always),
    // on request only (ICAP_MERGED=1): 152-165k against 176k captions/s on the side stream at
    // beam 5 - the record epilogue of these tiles and the attention both load the vector unit,
    // and with fc1 in front of the projection the side-stream attention is resident first
    if (c.precision == ICAP_PRECISION_BF16) {
      e->merged = merged_env && merged_env[0] == '1' && e->topk_able != 0 && c.W <= 5 &&
                  dense_topk_bf16_attq_supported(c.W, c.L, c.U, c.D, c.H, e->topk_able) &&
                  aligned16(weights->att_v_kernel);
      e->qtiles = e->merged;
    }
  }
  if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_lstm, hipEventDisableTiming);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_att, hipEventDisableTiming);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_gate, hipEventDisableTiming);
  if (err != hipSuccess) {
    if (e->ev_gate) (void)hipEventDestroy(e->ev_gate);
    if (e->ev_att) (void)hipEventDestroy(e->ev_att);
    if (e->ev_lstm) (void)hipEventDestroy(e->ev_lstm);
    if (e->ws) (void)hipFree(e->ws);
    delete e;
    return fail(ICAP_ERR_HIP, "icap_engine_create: %s", hipGetErrorString(err));
  }
  char* b = e->ws;
  e->features = (float*)(b + o_features); e->pf = (float*)(b + o_pf);
  e->pooled = (float*)(b + o_pooled); e->init_h = (float*)(b + o_init_h);
  e->init_c = (float*)(b + o_init_c); e->h[0] = (float*)(b + o_h0); e->h[1] = (float*)(b + o_h1);
  e->c = (float*)(b + o_c); e->q = (float*)(b + o_q); e->ctx = (float*)(b + o_ctx);
  e->fc1o = (float*)(b + o_fc1); e->logits = (float*)(b + o_logits); e->lp = (float*)(b + o_lp);
  e->hist_scores = (float*)(b + o_hs); e->hist_lp = (float*)(b + o_hl);
  e->len = (int32_t*)(b + o_len); e->hist_ids = (int32_t*)(b + o_hi);
  e->hist_parents = (int32_t*)(b + o_hp); e->start_ids = (int32_t*)(b + o_si);
  e->fin = (uint8_t*)(b + o_fin);
  e->beam_ws = b + o_bws;
  e->beam_ws_bytes = beam_step_workspace_bytes(c.B, c.W);
  e->emb_proj = (float*)(b + o_ep);
  e->row_len = (int32_t*)(b + o_rl);
  if (fold_able) {
    e->w12 = (float*)(b + o_w12);
    e->b12 = (float*)(b + o_b12);
    // default: exact-f32 mode.  Split mode keeps the two products unless asked to fold
    // (115k against 118k captions/s folded: its projection stays on the side-stream schedule,
    // and without fc1 in front of it the projection takes the compute units before the
    // attention arrives).
    const char* fold_env = getenv("ICAP_FOLD_FC");  // tuning knob, read per engine
    e->fold = fold_env ? fold_env[0] != '0' : c.precision == ICAP_PRECISION_FP32;
  }
  e->fin2 = (uint8_t*)(b + o_fin2);
  e->len2 = (int32_t*)(b + o_len2);
  e->flags = (int*)(b + o_flags);
  e->tile_rows_done = (unsigned*)(b + o_trd);
  e->q_done = (unsigned*)(b + o_qd);
  if (att_table) {
    if (!att16) e->epf = (float*)(b + o_epf);
    e->pf_extreme = (int*)(b + o_ext);
    if (att16) {
      e->epf16 = (uint16_t*)(b + o_epf16);
      e->feat16 = (uint16_t*)(b + o_f16);
    }
  }
  {
    // opt-in (ICAP_BEAM_IN_LAUNCH=1, read per engine): measured 7 % slower than the separate
    // beam-records kernel at the headline shape
    const char* bil_env = getenv("ICAP_BEAM_IN_LAUNCH");
    e->beam_in_launch =
        bil_env && bil_env[0] == '1' && c.precision == ICAP_PRECISION_FP32 && !e->wide &&
        e->topk_able != 0 &&
        dense_topk_beam_supported((int)N, c.W, dense_topk_half_tiles(c.V), e->topk_able, 1, 0);
  }
  e->fc1o16 = (uint16_t*)(b + o_fc116);
  {
    const char* sl_env = getenv("ICAP_SMALL_LOOP");  // tuning knob, read per engine
    e->small_loop = c.precision == ICAP_PRECISION_FP32 && c.cell == ICAP_CELL_LSTM && e->compute_units > 0 &&
                    small_loop_supported(c.B, c.W, c.D, c.H, c.U, c.L, c.D, c.V) &&
                    sl_env && sl_env[0] == '1';
  }
  e->split_cell = split_cell;
  if (split_cell) e->zgates = (float*)(b + o_zg);
  {
    const char* a16_env = getenv("ICAP_BF16_A16");  // tuning knob, read per engine
    e->a16 = !(a16_env && a16_env[0] == '0');
  }
  if (gru) {
    e->gru_k4 = (float*)(b + o_gk);
    e->gru_r4 = (float*)(b + o_gr);
    e->gru_b4 = (float*)(b + o_gb);
  }
  e->cell_kernel = gru ? e->gru_k4 : weights->lstm_kernel;
  e->cell_recurrent = gru ? e->gru_r4 : weights->lstm_recurrent_kernel;
  e->cell_bias = gru ? e->gru_b4 : weights->lstm_bias;
  e->rec = (float*)(b + o_rec);
  e->eos_x = (float*)(b + o_eosx);
  if (c.precision != ICAP_PRECISION_FP32)
    for (int i = 0; i < icap_engine::PK_COUNT; ++i) e->pk[i] = b + o_pk[i];
  if (c.precision == ICAP_PRECISION_F32_SPLIT) e->pk_fold = b + o_pkf;
  {
    const int rc = icap_engine_refresh_weights(e, nullptr);
    if (rc != ICAP_OK) {
      icap_engine_destroy(e);
      return rc;
    }
    const hipError_t se = hipStreamSynchronize(nullptr);
    if (se != hipSuccess) {
      icap_engine_destroy(e);
      return fail(ICAP_ERR_HIP, "icap_engine_create: %s", hipGetErrorString(se));
    }
  }
  *out = e;
  return ICAP_OK;
}

int icap_engine_refresh_weights(icap_engine* e, icap_stream_t stream) {
  ICAP_REQUIRE(e, "icap_engine_refresh_weights: null engine");
  const icap_config& c = e->cfg;
  const icap_weights& w = e->w;
  hipStream_t s = (hipStream_t)stream;
  const bool gru = c.cell == ICAP_CELL_GRU;
  if (gru)
    ICAP_TRY(gru_pack_f32(w.lstm_kernel, w.lstm_recurrent_kernel, w.lstm_bias, e->gru_k4, e->gru_r4,
                          e->gru_b4, c.D + c.E, c.H, s));
  const float* emb_rows = e->cell_kernel + (size_t)c.D * 4 * c.H;  // kernel rows of the embedding
  if (e->w12) {
    hipLaunchKernelGGL(fold_fc_kernel, dim3(ceil_div(c.V, 256), c.H + 1), dim3(256), 0, s, w.fc1_kernel,
                       w.fc1_bias, w.fc2_kernel, w.fc2_bias, e->w12, e->b12, c.H, c.V);
    ICAP_LAUNCH_CHECK("fold_fc_kernel");
  }
  if (c.precision != ICAP_PRECISION_FP32) {
    typedef icap_engine E;
    auto pack = c.precision == ICAP_PRECISION_BF16 ? pack_weight_bf16 : pack_weight_split;
    ICAP_TRY(pack(w.enc_kernel, e->pk[E::PK_ENC], c.F, c.D, 0, c.F, 0, s));
    ICAP_TRY(pack(w.att_w1_kernel, e->pk[E::PK_W1], c.D, c.U, 0, c.D, 0, s));
    ICAP_TRY(pack(w.init_h_kernel, e->pk[E::PK_INIT_H], c.D, c.H, 0, c.D, 0, s));
    if (!gru) ICAP_TRY(pack(w.init_c_kernel, e->pk[E::PK_INIT_C], c.D, c.H, 0, c.D, 0, s));
    ICAP_TRY(pack(w.att_w2_kernel, e->pk[E::PK_W2], c.H, c.U, 0, c.H, 0, s));
    // gate product over [ctx | h]: kernel rows [0,D) then the recurrent kernel, gates interleaved
    ICAP_TRY(pack(e->cell_kernel, e->pk[E::PK_LSTM], c.D, 4 * c.H, 0, c.D + c.H, c.H, s));
    ICAP_TRY(pack(e->cell_recurrent, e->pk[E::PK_LSTM], c.H, 4 * c.H, c.D, c.D + c.H, c.H, s));
    ICAP_TRY(pack(emb_rows, e->pk[E::PK_LSTM_EMB], c.E, 4 * c.H, 0, c.E, 0, s));
    ICAP_TRY(pack(w.fc1_kernel, e->pk[E::PK_FC1], c.H, c.H, 0, c.H, 0, s));
    ICAP_TRY(pack(w.fc2_kernel, e->pk[E::PK_FC2], c.H, c.V, 0, c.H, 0, s));
    if (e->w12) ICAP_TRY(pack(e->w12, e->pk_fold, c.H, c.V, 0, c.H, 0, s));
  }
  // emb_proj[v] = embedding[v] . lstm_kernel[D:D+E]: the part of the LSTM gates that
  // depends only on the input token, hoisted out of the step loop
  ICAP_TRY(engine_dense(e, icap_engine::PK_LSTM_EMB, w.embedding, emb_rows, nullptr, e->emb_proj, c.V,
                        c.E, 4 * c.H, ICAP_ACT_NONE, s));
  // the two scalar biases travel to the kernels by value: read them back (ordered behind
  // whatever wrote them on this stream; this is the one call that waits for its stream)
  ICAP_HIP(hipMemcpyAsync(&e->att_v_bias, w.att_v_bias, sizeof(float), hipMemcpyDeviceToHost, s));
  ICAP_HIP(hipMemcpyAsync(&e->fc3_bias, w.fc3_bias, sizeof(float), hipMemcpyDeviceToHost, s));
  ICAP_HIP(hipStreamSynchronize(s));
  return ICAP_OK;
}

int icap_engine_destroy(icap_engine* e) {
  if (!e) return ICAP_OK;
  for (auto& sp : e->spans) {
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  if (e->ev_gate) (void)hipEventDestroy(e->ev_gate);
  if (e->ev_att) (void)hipEventDestroy(e->ev_att);
  if (e->ev_lstm) (void)hipEventDestroy(e->ev_lstm);
  if (e->ws) (void)hipFree(e->ws);
  delete e;
  return ICAP_OK;
}

size_t icap_engine_workspace_bytes(const icap_engine* e) { return e ? e->ws_bytes : 0; }

int icap_decode(icap_engine* e, const float* img, int32_t* tokens, float* best_score,
                icap_stream_t stream) {
  ICAP_REQUIRE(e && img && tokens, "icap_decode: null argument");
  return decode_impl(e, img, tokens, best_score, (hipStream_t)stream);
}

int icap_search_baseline(icap_engine* e, const float* features, const int32_t* seq_len,
                         int pad_token, int32_t* tokens, float* best_score, icap_stream_t stream) {
  ICAP_REQUIRE(e && features && tokens, "icap_search_baseline: null argument");
  ICAP_REQUIRE(pad_token >= 0 && pad_token < e->cfg.V,
               "icap_search_baseline: pad token outside the vocabulary");
  ICAP_REQUIRE(!seq_len || (e->cfg.eos_id >= 0 && e->cfg.eos_id < e->cfg.V),
               "icap_search_baseline: forced eos needs an eos token inside the vocabulary");
  LoopVariant lv;
  lv.features = features;
  lv.seq_len = seq_len;
  lv.time_offset = 1;
  lv.mask_start = 1;
  lv.pad = pad_token;
  return decode_impl(e, nullptr, tokens, best_score, (hipStream_t)stream, lv);
}

int icap_logit_masks_f32(float* logits, int N, int V, int start_token, const int32_t* row_len,
                         int step_plus_1, int eos_token, int pad_token, icap_stream_t stream) {
  LogitMasks mk;
  mk.start_col = start_token;
  mk.row_len = row_len;
  mk.step1 = step_plus_1;
  mk.eos = eos_token;
  mk.pad = pad_token;
  return logit_masks_f32(logits, N, V, mk, (hipStream_t)stream);
}

int icap_engine_trace(const icap_engine* e, int32_t* ids, int32_t* parents, float* scores,
                      float* log_probs, icap_stream_t stream) {
  ICAP_REQUIRE(e, "icap_engine_trace: null engine");
  const size_t n = (size_t)e->cfg.T * e->cfg.B * e->cfg.W * 4;
  hipStream_t s = (hipStream_t)stream;
  if (ids) ICAP_HIP(hipMemcpyAsync(ids, e->hist_ids, n, hipMemcpyDeviceToDevice, s));
  if (parents) ICAP_HIP(hipMemcpyAsync(parents, e->hist_parents, n, hipMemcpyDeviceToDevice, s));
  if (scores) ICAP_HIP(hipMemcpyAsync(scores, e->hist_scores, n, hipMemcpyDeviceToDevice, s));
  if (log_probs) ICAP_HIP(hipMemcpyAsync(log_probs, e->hist_lp, n, hipMemcpyDeviceToDevice, s));
  return ICAP_OK;
}

int icap_engine_set_timing(icap_engine* e, unsigned class_mask) {
  ICAP_REQUIRE(e, "icap_engine_set_timing: null engine");
  if (class_mask) {
    // launches a decode can bracket: a few once-per-decode classes plus at most seven per step;
    // the pool serves ICAP_TIMING_POOL_DECODES bracketed decodes between two reads
    const size_t want = (size_t)ICAP_TIMING_POOL_DECODES * (8 + 7 * (size_t)e->cfg.T);
    while (e->spans.size() < want) {
      icap_engine::Span sp;
      sp.cls = 0;
      sp.a = sp.b = nullptr;
      if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) {
        if (sp.a) (void)hipEventDestroy(sp.a);
        return fail(ICAP_ERR_HIP, "icap_engine_set_timing: cannot create the timing events");
      }
      e->spans.push_back(sp);
    }
  }
  e->timing_mask = class_mask;
  e->timing = class_mask;
  e->decode_calls = 0;
  e->spans_used = 0;
  e->spans_dropped = 0;
  return ICAP_OK;
}

int icap_engine_set_timing_every(icap_engine* e, int n) {
  ICAP_REQUIRE(e, "icap_engine_set_timing_every: null engine");
  ICAP_REQUIRE(n >= 1, "icap_engine_set_timing_every: n must be at least 1 (got %d)", n);
  e->timing_every = n;
  e->decode_calls = 0;
  return ICAP_OK;
}

int icap_engine_set_fused_vocab(icap_engine* e, int on) {
  ICAP_REQUIRE(e, "icap_engine_set_fused_vocab: null engine");
  if (on && !e->topk_able)
    return fail(ICAP_ERR_UNSUPPORTED,
                "icap_engine_set_fused_vocab: this engine's shape or precision is not served by "
                "the fused projection (fp32: vocab %% 4 == 0, H %% 32 == 0; beam width <= 5 in the "
                "bf16 and split modes)");
  e->topk = on ? e->topk_able : 0;
  return ICAP_OK;
}

int icap_engine_fused_vocab(const icap_engine* e) { return e && e->topk ? 1 : 0; }

int icap_engine_set_overlap(icap_engine* e, int on) {
  ICAP_REQUIRE(e, "icap_engine_set_overlap: null engine");
  e->overlap = on != 0;
  return ICAP_OK;
}

int icap_engine_merged_launch(const icap_engine* e) {
  return e && e->merged && e->overlap && e->topk != 0 ? 1 : 0;
}

int icap_engine_fold_fc(const icap_engine* e) { return e && e->fold && e->w12 ? 1 : 0; }

int icap_engine_att_table(const icap_engine* e) { return e && (e->epf || e->epf16) ? 1 : 0; }

int icap_engine_small_loop(const icap_engine* e) { return e && e->small_loop ? 1 : 0; }

int icap_engine_decode_status(icap_engine* e, int32_t* invalid, icap_stream_t stream) {
  ICAP_REQUIRE(e && invalid, "icap_engine_decode_status: null argument");
  *invalid = 0;
  hipStream_t s = (hipStream_t)stream;
  if (e->last_err) {
    int word = 0;
    ICAP_HIP(hipMemcpyAsync(&word, e->last_err, sizeof(int), hipMemcpyDeviceToHost, s));
    ICAP_HIP(hipStreamSynchronize(s));
    *invalid = word != 0;
  } else {
    ICAP_HIP(hipStreamSynchronize(s));
  }
  return ICAP_OK;
}

int icap_engine_debug_sync_state(const icap_engine* e, icap_stream_t stream, int32_t state[4]) {
  ICAP_REQUIRE(e && state, "icap_engine_debug_sync_state: null argument");
  state[0] = (int32_t)hipStreamQuery((hipStream_t)stream);
  state[1] = (int32_t)hipStreamQuery(e->side);
  state[2] = (int32_t)hipEventQuery(e->ev_lstm);
  state[3] = (int32_t)hipEventQuery(e->ev_att);
  (void)hipGetLastError();  // hipErrorNotReady is an answer here, not an error to keep
  return ICAP_OK;
}

int icap_engine_set_fold_fc(icap_engine* e, int on) {
  ICAP_REQUIRE(e, "icap_engine_set_fold_fc: null engine");
  if (on && !e->w12)
    return fail(ICAP_ERR_UNSUPPORTED, "icap_engine_set_fold_fc: the folded output kernel serves fp32 and split mode");
  e->fold = on != 0;
  return ICAP_OK;
}

int icap_engine_kernel_times(icap_engine* e, float* ms, int32_t* launches) {
  ICAP_REQUIRE(e && ms && launches, "icap_engine_kernel_times: null argument");
  for (int i = 0; i < ICAP_K_COUNT; ++i) {
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
