// lf_api.hip -- host side of the C ABI declared in include/linefront.h.
//
// Owns the context (HIP stream, device buffers sized for 288 GB HBM3E: every per-frame product of
// a whole batch stays resident), builds the small host tables the reference computes with libm on
// the CPU (Gaussian taps lsd.cpp:461-487, log-gamma lsd.cpp:886-934, log p), and launches the
// kernels.  There is no CPU compute path in this file.
#include "../../include/linefront.h"
#include "lf_lsd.h"
#include "lf_front.h"
#include "lf_pair.h"
#include "lf_points.h"
#include "lf_orb.h"
#include "lf_edlines.h"
#include "lf_pair_legacy.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is bound at run time (dlopen), never at link time
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define LF_VERSION_STR "linefront-mi355x 0.1 (gfx950)"

struct lf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int W = 0, H = 0, maxB = 0;
  lf_params params;
  lf_caps caps;
  LsdConsts lc;
  LsdBuffers lb;
  std::vector<void *> allocs;
  struct Guard { void *p; size_t bytes; const char *what; };
  std::vector<Guard> guards;       // LF_DEBUG_GUARD=1: tails to check at destruction
  FrontConsts fc;
  FrontBuffers fb;
  uint64_t *d_frame_ids = nullptr;
  PairConsts pcn;
  PairBuffers pb;
  int *d_pair_q = nullptr, *d_pair_t = nullptr;
  int last_pairs = 0;
  // pinned staging for the small host arrays of the batched entry points (frame ids, pair lists): the calls
  // stay asynchronous, so that several contexts on different streams can be driven from one host thread
  uint8_t *h_stage = nullptr;
#define LF_PAIR_STAGE_SLOTS 4   // pinned staging slots of the pair lists: several pair launches per pass (odometry + loop
                                // closures, feature matching + hybrid solve) must not make the host wait for the pass
  hipEvent_t ev_stage_ids = nullptr, ev_stage_pairs[LF_PAIR_STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  bool stage_ids_pending = false, stage_pairs_pending[LF_PAIR_STAGE_SLOTS] = {false, false, false, false};
  unsigned stage_pairs_next = 0;
  bool hybrid_ready = false;         // hybrid (points + lines) buffers are allocated on first use
  int *d_pm_q = nullptr, *d_pm_t = nullptr, *d_npm = nullptr;
  float *d_pts_stage = nullptr;      // lf_match_node_pair_hybrid staging: 2 x LF_NODE_PT_CAP float4
  int *d_legacy_i = nullptr;         // lf_relative_transformation_legacy staging: match q | t | inlier indices (3 x LF_LEGACY_CAP)
  float *d_legacy_f = nullptr;       //   match distances
  LegacyResult *d_legacy_r = nullptr;
  // the point front end (ORB extraction, projectTo3D) on its own stream, next to the line front end (lf_ctx_point_stream)
  hipStream_t pstream = nullptr;
  hipEvent_t ev_pts_in = nullptr, ev_pts_done = nullptr, ev_pts_free = nullptr;
  bool pts_async = false, pts_pending = false, pts_free_rec = false;
  bool orb_adj_set = false; double h_orb_adj = 0.0;
  int *d_orb_thr = nullptr;          // [maxB] per-frame FAST thresholds of the adjusted extractor
  double *d_orb_adj = nullptr;       // [1] DetectorAdjuster::thresh_ (device-resident state, carried from call to call)
  uint8_t *d_fm_stage = nullptr;     // lf_feature_match_node_pair staging: 2 x 1024 descriptors of 32 bytes (first use)
  int32_t *d_fm_n = nullptr, *d_fm_q = nullptr, *d_fm_t = nullptr, *d_fm_cnt = nullptr;
  float *d_fm_d = nullptr;
  bool last_hybrid = false;
  PairBuffers last_pb;               // the buffers of the last pair launch (train side may be an external map)
  unsigned char *d_adjacent = nullptr;   // [maxB] adjacentFrame flags of lf_line_matching_device
  double *d_descdiff = nullptr;      // lf_pair_get_descdiff scratch (line_cap^2 doubles), allocated on first use
  // ---- EDLines (buffers allocated on first use)
  bool ed_ready = false;
  std::vector<int> h_xslots;    // the key-frame slot list last uploaded for the exchange
  EdConsts ec;
  EdBuffers eb;
  // ---- ORB extractor (buffers allocated on first use)
  bool orb_ready = false;
  OrbConsts oc;
  OrbBuffers ob;
  int *d_orb_nkp = nullptr;
  int orb_last = 0;
  // ---- key-frame exchange over RCCL
  ncclComm_t comm = nullptr;
  long long n_allgathers = 0;       // ncclAllGather calls this context has issued (lf_comm_info)
  size_t alloc_bytes = 0;           // device memory this context has allocated (lf_ctx_device_bytes)
  bool comm_owner = false;
  int comm_world = 0, comm_rank = 0, comm_max_kf = 0;
  int xbuf_world = 0, xbuf_max_kf = 0;      // geometry the exchange buffers were allocated for
  lf_line_record *d_xsend = nullptr, *d_xrecv = nullptr;   // [max_kf][line_cap + 1], [world * max_kf][line_cap + 1]
  int *d_xnlines = nullptr, *d_xslots = nullptr;
  uint64_t *d_xids = nullptr;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint8_t *d_gray_stage = nullptr;   // staging for the host-pointer convenience entry points
  float *d_depth_stage = nullptr;
  int last_batch = 0;
  std::string err;
};

// ------------------------------------------------------------------------------------------------
static int fail_hip(lf_ctx *c, hipError_t e, const char *what) {
  if (c) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    c->err = buf;
  }
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu ||
          e == hipErrorInsufficientDriver)
             ? LF_ERR_NO_DEVICE
             : LF_ERR_HIP;
}
#define HIPCHK(ctx, call)                                   \
  do {                                                      \
    hipError_t e_ = (call);                                 \
    if (e_ != hipSuccess) return fail_hip(ctx, e_, #call);  \
  } while (0)

// LF_DEBUG_GUARD=1 in the environment: every device allocation gets a 4 KB tail filled with 0xA5, checked when the context is
// destroyed (a kernel that writes past the end of its buffer is named on stderr with the buffer's expression and size)
#define LF_GUARD_BYTES 4096
static bool guard_on() { static int g = -1; if (g < 0) { const char *e = getenv("LF_DEBUG_GUARD"); g = (e && e[0] == '1') ? 1 : 0; } return g == 1; }
template <typename T>
static int dev_alloc(lf_ctx *c, T **p, size_t count, const char *what) {
  void *q = nullptr;
  const size_t bytes = count * sizeof(T);
  hipError_t e = hipMalloc(&q, bytes + 256 + (guard_on() ? LF_GUARD_BYTES : 0));
  if (e != hipSuccess) {
    if (e == hipErrorOutOfMemory) {       // name the buffer and the device's state: a batch sized past the HBM is a capacity error
      size_t fr = 0, tot = 0;
      (void)hipGetLastError();
      (void)hipMemGetInfo(&fr, &tot);
      char buf[320];
      snprintf(buf, sizeof buf, "device memory: %s needs %.2f GB, %.2f of %.2f GB free (this context holds %.2f GB already)", what,
               bytes / 1e9, fr / 1e9, tot / 1e9, c->alloc_bytes / 1e9);
      c->err = buf;
      return LF_ERR_CAPACITY;
    }
    return fail_hip(c, e, "hipMalloc");
  }
  c->alloc_bytes += bytes;
  c->allocs.push_back(q);
  if (guard_on()) {
    (void)hipMemset((char *)q + bytes, 0xA5, 256 + LF_GUARD_BYTES);
    c->guards.push_back({q, bytes, what});
  }
  *p = reinterpret_cast<T *>(q);
  return LF_OK;
}
#define ALLOC(ctx, ptr, count)                                 \
  do {                                                         \
    int r_ = dev_alloc(ctx, &(ptr), (size_t)(count), #ptr);    \
    if (r_ != LF_OK) return r_;                                \
  } while (0)
static void guard_check_one(const lf_ctx::Guard &g) {
  std::vector<unsigned char> h(256 + LF_GUARD_BYTES);
  if (hipMemcpy(h.data(), (const char *)g.p + g.bytes, h.size(), hipMemcpyDeviceToHost) != hipSuccess) return;
  for (size_t i = 0; i < h.size(); i++)
    if (h[i] != 0xA5) { fprintf(stderr, "linefront guard: %s (%zu bytes) overwritten at +%zu past its end\n", g.what, g.bytes, i); break; }
}
static void guard_check(lf_ctx *c) {
  for (const auto &g : c->guards) guard_check_one(g);
}

static void pt_stream_join(lf_ctx *c);
static void pt_stream_consumed(lf_ctx *c);

extern "C" {

void lf_params_init(lf_params *p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->lsd_angle_th = 22.5;            // parameter_server.cpp:162
  p->lsd_density_th = 0.7;           // :163
  p->lsd_scale = 0.8;                // lsd.cpp:2097
  p->lsd_sigma_scale = 0.6;          // lsd.cpp:2073
  p->lsd_quant = 2.0;                // lsd.cpp:2075
  p->lsd_log_eps = 0.0;              // lsd.cpp:2078
  p->lsd_n_bins = 1024;              // lsd.cpp:2080
  p->lsd_max_grad = 255.0;           // lsd.cpp:2082
  p->line_segment_len_thresh = 10;   // :164
  p->line3d_length_thresh = 0.02;    // :169
  p->ratio_of_collinear_pts = 0.6;   // :170
  p->line_sample_max_num = 100;      // :171
  p->line_sample_min_num = 10;       // :172
  p->line_sample_interval = 1;       // :173
  p->line3d_mle_iter_num = 100;      // :174
  p->pt2line_mahdist_extractline = 1.5;   // :176
  p->ransac_iters_extract_line = 100;     // :177
  p->num_cells_lineseg_range = 10;        // :178
  p->ratio_support_pts_on_line = 0.7;     // :179
  p->stdev_sample_pt_imgline = 3;         // :182
  p->depth_stdev_coeff_c1 = 0.00273;      // :183
  p->depth_stdev_coeff_c2 = 0.00074;      // :184
  p->depth_stdev_coeff_c3 = -0.00058;     // :185
  p->msld_sample_interval = 1;            // :166
  p->depth_scaling = 1.0;                 // node.cpp:214
  p->ransac_iters_line_motion = 500;      // :190
  p->adjacent_linematch_window = 3;       // :191
  p->line_match_number_weight = 1;        // :197
  p->min_feature_matches = 20;            // parameter_server.cpp:82
  p->min_matches_loopclose = 20;          // :198
  p->max_mah_dist_for_inliers = 3;        // :193
  p->g2o_line_error_weight = 1;           // :194
  p->g2o_BA_use_kernel = 1;               // lineslam.cpp:629
  p->g2o_BA_kernel_delta = 10;            // lineslam.cpp:630
  p->rng_seed = 0;
  p->pt2line3d_dist_relmotion = 0.05;     // :188
  p->line3d_angle_relmotion = 10;         // :189
  p->line_detector = LF_DETECTOR_LSD;     // Node::Node passes "LSD" (src/node.cpp:214)
}
void lf_caps_init(lf_caps *k) {
  if (!k) return;
  k->seg_cap = 4096; k->line_cap = LF_MATCH_LINE_CAP; k->match_cap = LF_MAX_MATCHES; k->pt_match_cap = LF_MAX_PT_MATCHES;
}
void lf_params_init_launch(lf_params *p) {
  lf_params_init(p);
  if (!p) return;
  p->lsd_angle_th = 40;          // launch/lineslam.launch:39
  p->min_feature_matches = 10;   // launch/lineslam.launch (min_matches)
}
const char *lf_version(void) { return LF_VERSION_STR; }
const char *lf_status_str(int s) {
  switch (s) {
    case LF_OK: return "ok";
    case LF_ERR_INVALID: return "invalid argument";
    case LF_ERR_NO_DEVICE: return "no usable gfx950 HIP device (this library has no CPU path)";
    case LF_ERR_HIP: return "HIP runtime error";
    case LF_ERR_CAPACITY: return "buffer capacity exceeded";
    case LF_ERR_UNSUPPORTED: return "unsupported parameter value for this build";
    default: return "unknown status";
  }
}
// lf_last_error(NULL): the message of the last lf_ctx_create* that FAILED on this thread (its context is gone with its message)
static thread_local std::string g_create_err;
const char *lf_last_error(const lf_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }
}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Host tables.  These are the quantities the reference evaluates with the host libm; they depend
// only on the image size and the parameters, not on pixel data.
static double host_log_gamma(double x) {   // lsd.cpp:886-934
  if (x > 15.0)
    return 0.918938533204673 + (x - 0.5) * log(x) - x +
           0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705,
                              1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * log(x + 5.5) - (x + 5.5), b = 0.0;
  for (int n = 0; n < 7; n++) {
    a -= log(x + (double)n);
    b += q[n] * pow(x, (double)n);
  }
  return a + log(b);
}
static void host_gauss_taps(double sigma, double mean, int n, double *k) {   // lsd.cpp:461-487
  double sum = 0.0;
  for (int i = 0; i < n; i++) {
    double val = ((double)i - mean) / sigma;
    k[i] = exp(-0.5 * val * val);
    sum += k[i];
  }
  if (sum >= 0.0)
    for (int i = 0; i < n; i++) k[i] /= sum;
}

static int build_lsd_consts(lf_ctx *c) {
  const lf_params &p = c->params;
  LsdConsts &lc = c->lc;
  memset(&lc, 0, sizeof lc);
  if (!(p.lsd_scale > 0.0) || !(p.lsd_sigma_scale > 0.0) || p.lsd_quant < 0.0 ||
      !(p.lsd_angle_th > 0.0 && p.lsd_angle_th < 180.0) || p.lsd_density_th < 0.0 ||
      p.lsd_density_th > 1.0 || p.lsd_n_bins <= 0 || p.lsd_n_bins > 1024 || !(p.lsd_max_grad > 0.0))
    return LF_ERR_INVALID;   // the checks of lsd.cpp:1949-1960 (reference: exit())
  lc.W = c->W; lc.H = c->H;
  lc.scale = p.lsd_scale;
  if (p.lsd_scale != 1.0) {
    lc.N = (int)(unsigned int)floor(c->W * p.lsd_scale);   // lsd.cpp:549-550
    lc.M = (int)(unsigned int)floor(c->H * p.lsd_scale);
  } else { lc.N = c->W; lc.M = c->H; }
  if (lc.N < 2 || lc.M < 2 || lc.N > 65535 || lc.M > 65535) return LF_ERR_INVALID;
  double sigma = p.lsd_scale < 1.0 ? p.lsd_sigma_scale / p.lsd_scale : p.lsd_sigma_scale;
  int h = (int)(unsigned int)ceil(sigma * sqrt(2.0 * 3.0 * log(10.0)));   // lsd.cpp:564-565
  lc.ntaps = 1 + 2 * h;
  lc.n_bins = p.lsd_n_bins;
  lc.max_grad = p.lsd_max_grad;
  lc.prec = M_PI * p.lsd_angle_th / 180.0;                  // lsd.cpp:1963
  lc.p = p.lsd_angle_th / 180.0;                            // lsd.cpp:1964
  lc.rho = p.lsd_quant / sin(lc.prec);                      // lsd.cpp:1965
  lc.cos_prec = cos(lc.prec);
  lc.k_hi = lc.cos_prec * lc.cos_prec + 1e-12; lc.k_lo = lc.cos_prec * lc.cos_prec - 1e-12;
  lc.logNT = 5.0 * (log10((double)lc.N) + log10((double)lc.M)) / 2.0;   // lsd.cpp:1983
  lc.min_reg_size = (int)(-lc.logNT / log10(lc.p));         // lsd.cpp:1984
  lc.density_th = p.lsd_density_th;
  lc.eps = p.lsd_log_eps;
  double pk = lc.p;
  for (int k = 0; k < LF_MAX_PLEVEL; k++) {   // rect_improve halves p (lsd.cpp:1677,1755)
    lc.logp[k] = log(pk);
    lc.log1mp[k] = log(1.0 - pk);
    lc.log10p[k] = log10(pk);
    pk /= 2.0;
  }
  lc.seg_cap = c->caps.seg_cap;
  {
    // wavefronts per frame in the seed sweep: 0 = automatic (speculative 8-wave sweep for small batches,
    // where per-frame latency is what matters; sequential 1-wave sweep once >= 1 frame per SIMD is in flight)
    const char *e = getenv("LF_SWEEP_WAVES");
    lc.sweep_waves = e ? atoi(e) : 0;
    if (lc.sweep_waves < 0) lc.sweep_waves = 0;
    if (lc.sweep_waves > LF_MW_MAXW) lc.sweep_waves = LF_MW_MAXW;
    // LF_SWEEP_LU=1: the one-wavefront sweep keeps `used` (+ NOTDEF) as a bitmap in LDS and stages the seeds' (cos, sin) tiles in
    // LDS by DMA (k_lsd_sweep_lu, 30 KB of LDS per frame).  Bit-identical; measured: a pass alone 105.5 ->
    // 98.9 ms, but five frames fill a CU's LDS, so the passes of a pipelined run no longer share CUs with each other or with
    // k_mle: four passes in flight 119 -> 153 ms.  Off by default.
    const char *lu = getenv("LF_SWEEP_LU");
    lc.sweep_lu = lu ? atoi(lu) : 0;
  }
  return LF_OK;
}

static int upload_lsd_tables(lf_ctx *c) {
  const lf_params &p = c->params;
  const LsdConsts &lc = c->lc;
  const int n = lc.ntaps, h = (n - 1) / 2;
  double sigma = p.lsd_scale < 1.0 ? p.lsd_sigma_scale / p.lsd_scale : p.lsd_sigma_scale;
  std::vector<double> kx((size_t)lc.N * n), ky((size_t)lc.M * n);
  std::vector<int> jx((size_t)lc.N * n), jy((size_t)lc.M * n);
  for (int x = 0; x < lc.N; x++) {   // lsd.cpp:573-604
    double xx = (double)x / p.lsd_scale;
    int xc = (int)floor(xx + 0.5);
    host_gauss_taps(sigma, (double)h + xx - (double)xc, n, &kx[(size_t)x * n]);
    for (int i = 0; i < n; i++) {
      int j = xc - h + i, d = 2 * lc.W;
      while (j < 0) j += d;
      while (j >= d) j -= d;
      if (j >= lc.W) j = d - 1 - j;
      jx[(size_t)x * n + i] = j;
    }
  }
  for (int y = 0; y < lc.M; y++) {   // lsd.cpp:607-638
    double yy = (double)y / p.lsd_scale;
    int yc = (int)floor(yy + 0.5);
    host_gauss_taps(sigma, (double)h + yy - (double)yc, n, &ky[(size_t)y * n]);
    for (int i = 0; i < n; i++) {
      int j = yc - h + i, d = 2 * lc.H;
      while (j < 0) j += d;
      while (j >= d) j -= d;
      if (j >= lc.H) j = d - 1 - j;
      jy[(size_t)y * n + i] = j;
    }
  }
  if (p.lsd_scale == 1.0) {   // no sampling: identity taps (lsd.cpp:1968-1978 skips the sampler)
    for (int x = 0; x < lc.N; x++) for (int i = 0; i < n; i++) { kx[(size_t)x * n + i] = (i == h); jx[(size_t)x * n + i] = x; }
    for (int y = 0; y < lc.M; y++) for (int i = 0; i < n; i++) { ky[(size_t)y * n + i] = (i == h); jy[(size_t)y * n + i] = y; }
  }
  const size_t NM = (size_t)lc.N * lc.M;
  // (k_nfa_table tabulates n < LF_NFA_TAB_N whatever the image size: the table of log-gammas covers that as well)
  std::vector<double> lg(NM + 2 > (size_t)LF_NFA_TAB_N + 2 ? NM + 2 : (size_t)LF_NFA_TAB_N + 2);
  lg[0] = 0.0;
  for (size_t i = 1; i < lg.size(); i++) lg[i] = host_log_gamma((double)i);
  HIPCHK(c, hipMemcpyAsync((void *)c->lb.kx, kx.data(), kx.size() * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((void *)c->lb.ky, ky.data(), ky.size() * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((void *)c->lb.jx, jx.data(), jx.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((void *)c->lb.jy, jy.data(), jy.size() * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((void *)c->lb.lgam, lg.data(), lg.size() * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync((void *)c->lb.dconsts, &c->lc, sizeof(LsdConsts), hipMemcpyHostToDevice, c->stream));
  lf_lsd_build_tables(c->lc, c->lb, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));   // host vectors go out of scope
  return LF_OK;
}

static int alloc_lsd(lf_ctx *c) {
  const LsdConsts &lc = c->lc;
  LsdBuffers &b = c->lb;
  memset(&b, 0, sizeof b);
  const size_t B = (size_t)c->maxB, NM = (size_t)lc.N * lc.M;
  // size tables for the worst case over parameter changes of the same context (scale fixed)
  double *kx, *ky, *lgam; int *jx, *jy; LsdConsts *dc;
  ALLOC(c, kx, (size_t)lc.N * lc.ntaps); ALLOC(c, ky, (size_t)lc.M * lc.ntaps);
  ALLOC(c, jx, (size_t)lc.N * lc.ntaps); ALLOC(c, jy, (size_t)lc.M * lc.ntaps);
  ALLOC(c, lgam, NM + 2 > (size_t)LF_NFA_TAB_N + 2 ? NM + 2 : (size_t)LF_NFA_TAB_N + 2); ALLOC(c, dc, 1);
  b.kx = kx; b.ky = ky; b.jx = jx; b.jy = jy; b.lgam = lgam; b.dconsts = dc;
  {
    const char *e = getenv("LF_NFA_TABLE");   // 0: evaluate nfa() every time (A/B parity test of the table)
    if (!e || atoi(e) != 0) ALLOC(c, b.nfa_tab, (size_t)LF_MAX_PLEVEL * LF_NFA_TAB_TRI);
  }
  ALLOC(c, b.aux, B * lc.H * lc.N);
  ALLOC(c, b.scaled, B * NM);
  ALLOC(c, b.angles, B * NM);
  ALLOC(c, b.modgrad, B * NM);
  ALLOC(c, b.cossin, B * NM * 2);
  ALLOC(c, b.bins, B * NM);
  int nch = (lc.N - 1 + LF_SORT_CHUNK_COLS - 1) / LF_SORT_CHUNK_COLS;
  ALLOC(c, b.cnt, B * nch * 1024);
  ALLOC(c, b.seeds, B * NM);
  ALLOC(c, b.nseeds, B);
  ALLOC(c, b.used, B * NM);
  if (lc.sweep_lu) ALLOC(c, b.ndbits, B * (size_t)lc.M * (size_t)((lc.N + 31) / 32));      // k_lsd_sweep_lu's initial bitmap (opt-in)
  ALLOC(c, b.reg, B * NM);
  ALLOC(c, b.tmp, B * NM);
  ALLOC(c, b.mw_tag, B * LF_MW_MAXW * NM);
  ALLOC(c, b.mw_lists, B * LF_MW_MAXW * 4 * (size_t)LF_MW_CAP);
  HIPCHK(c, hipMemsetAsync(b.mw_tag, 0, B * LF_MW_MAXW * NM, c->stream));
  ALLOC(c, b.labels, B * NM);
  ALLOC(c, b.segs, B * (size_t)lc.seg_cap * LF_SEG_STRIDE);
  ALLOC(c, b.nsegs, B);
  ALLOC(c, b.stats, B * LF_STATS_STRIDE);
  ALLOC(c, c->d_gray_stage, (size_t)c->W * c->H);
  ALLOC(c, c->d_depth_stage, (size_t)c->W * c->H);
  // ---- 3D-line stage
  FrontConsts &fc = c->fc;
  FrontBuffers &fb = c->fb;
  memset(&fc, 0, sizeof fc);
  memset(&fb, 0, sizeof fb);
  fc.W = c->W; fc.H = c->H;
  fc.cand_cap = lc.seg_cap; fc.line_cap = c->caps.line_cap; fc.seg_cap = lc.seg_cap;   // every LSD segment is examined
  fc.pts_rows = 0;
  ALLOC(c, fb.gxy, B * (size_t)c->W * c->H * 2);
  ALLOC(c, c->d_frame_ids, B);
  ALLOC(c, fb.cand_flag, B * fc.cand_cap);
  ALLOC(c, fb.cand_out, B * (size_t)fc.cand_cap * LF_CAND_STRIDE);
  ALLOC(c, fb.cand_mask, B * (size_t)fc.cand_cap * 2);
  ALLOC(c, fb.pts, (size_t)fc.line_cap * LF_MAX_SAMPLES * 3);      // lf_mle_lines' staging only
  ALLOC(c, fb.recs, B * (size_t)fc.line_cap);
  ALLOC(c, fb.nlines, B);
  ALLOC(c, fb.mle_list, B * 3 * (size_t)fc.line_cap);
  ALLOC(c, fb.mle_cnt, B * 3);
  fb.frame_ids = c->d_frame_ids;
  fb.segs = b.segs; fb.nsegs = b.nsegs;
  // ---- pair solver (at most one pair per frame slot)
  PairConsts &pcn = c->pcn;
  PairBuffers &pb = c->pb;
  memset(&pcn, 0, sizeof pcn);
  memset(&pb, 0, sizeof pb);
  pcn.line_cap = fc.line_cap; pcn.match_cap = c->caps.match_cap; pcn.pt_match_cap = c->caps.pt_match_cap;
  pcn.mode = LF_MODE_SOLVE; pcn.refine_iters = 0;
  pcn.cos_angle_thresh = cos(30 * 3.14159265 / 180);   // node.cpp:1624 with lineslam.h:38 PI
  pcn.cos_degeneracy = cos(5 * 3.14159265 / 180);      // motion.cpp:407
  ALLOC(c, c->d_pair_q, B); ALLOC(c, c->d_pair_t, B);
  ALLOC(c, pb.live_idx, B * (size_t)fc.line_cap * fc.line_cap);
  ALLOC(c, pb.live_val, B * (size_t)fc.line_cap * fc.line_cap);
  ALLOC(c, c->d_adjacent, B);
  pb.adjacent = nullptr;
  ALLOC(c, pb.match_q, B * (size_t)pcn.match_cap); ALLOC(c, pb.match_t, B * (size_t)pcn.match_cap);
  ALLOC(c, pb.match_d, B * (size_t)pcn.match_cap);
  ALLOC(c, pb.nmatches, B);
  ALLOC(c, pb.results, B);
  ALLOC(c, pb.inliers, B * (size_t)LF_MAX_MATCHES);
  ALLOC(c, pb.ws, B * (size_t)LF_PAIR_WS_DOUBLES);
  ALLOC(c, pb.motion_d, B * (size_t)LF_MOTION_STRIDE);
  pb.recs = fb.recs; pb.nlines = fb.nlines; pb.frame_ids = c->d_frame_ids;
  pb.recs_t = fb.recs; pb.nlines_t = fb.nlines; pb.frame_ids_t = c->d_frame_ids; pb.line_cap_t = fc.line_cap;
  pb.pair_q = c->d_pair_q; pb.pair_t = c->d_pair_t;
  return LF_OK;
}

extern "C" {

int lf_ctx_create(lf_ctx **out, int device, void *hip_stream, int width, int height, int max_batch,
                  const lf_params *params) {
  return lf_ctx_create_caps(out, device, hip_stream, width, height, max_batch, params, nullptr);
}
int lf_ctx_get_caps(const lf_ctx *c, lf_caps *k) {
  if (!c || !k) return LF_ERR_INVALID;
  *k = c->caps;
  return LF_OK;
}
int lf_ctx_create_caps(lf_ctx **out, int device, void *hip_stream, int width, int height, int max_batch,
                       const lf_params *params, const lf_caps *caps) {
  if (!out || width < 8 || height < 8 || max_batch < 1) return LF_ERR_INVALID;
  *out = nullptr;
  lf_caps k;
  lf_caps_init(&k);
  if (caps) {
    if (caps->seg_cap < 1 || caps->line_cap < 1 || caps->match_cap < 3 || caps->pt_match_cap < 0) return LF_ERR_INVALID;
    if (caps->seg_cap > 65535 /* 16-bit region labels */ || caps->line_cap > LF_MATCH_LINE_CAP || caps->match_cap > LF_MAX_MATCHES ||
        caps->pt_match_cap > LF_MAX_PT_MATCHES)
      return LF_ERR_UNSUPPORTED;
    k = *caps;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return LF_ERR_NO_DEVICE;
  lf_ctx *c = new lf_ctx();
  c->device = device;
  c->W = width; c->H = height; c->maxB = max_batch;
  c->caps = k;
  if (params) c->params = *params; else lf_params_init(&c->params);
  int r = LF_OK;
  do {
    if ((e = hipSetDevice(device)) != hipSuccess) { r = fail_hip(c, e, "hipSetDevice"); break; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
      if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { r = fail_hip(c, e, "hipStreamCreate"); break; }
      c->own_stream = true;
    }
    for (int i = 0; i < 8; i++) if ((e = hipEventCreate(&c->ev[i])) != hipSuccess) { r = fail_hip(c, e, "hipEventCreate"); break; }
    if (r != LF_OK) break;
    e = hipEventCreateWithFlags(&c->ev_stage_ids, hipEventDisableTiming);
    for (int i = 0; i < LF_PAIR_STAGE_SLOTS && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->ev_stage_pairs[i], hipEventDisableTiming);
    if (e != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_stage, (size_t)c->maxB * 8 * (1 + LF_PAIR_STAGE_SLOTS), hipHostMallocDefault)) != hipSuccess) {
      r = fail_hip(c, e, "staging buffers");
      break;
    }
    if (r != LF_OK) break;
    if ((r = build_lsd_consts(c)) != LF_OK) break;
    if ((r = alloc_lsd(c)) != LF_OK) break;
    if ((r = upload_lsd_tables(c)) != LF_OK) break;
  } while (0);
  if (r != LF_OK) { g_create_err = c->err; lf_ctx_destroy(c); return r; }
  g_create_err.clear();
  *out = c;
  return LF_OK;
}

void lf_ctx_destroy(lf_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)lf_comm_destroy(c);
  if (c->pstream) {
    (void)hipStreamSynchronize(c->pstream);
    (void)hipEventDestroy(c->ev_pts_in); (void)hipEventDestroy(c->ev_pts_done); (void)hipEventDestroy(c->ev_pts_free);
    (void)hipStreamDestroy(c->pstream);
  }
  if (!c->guards.empty()) guard_check(c);
  for (void *p : c->allocs) (void)hipFree(p);
  for (int i = 0; i < 8; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->ev_stage_ids) (void)hipEventDestroy(c->ev_stage_ids);
  for (int i = 0; i < LF_PAIR_STAGE_SLOTS; i++) if (c->ev_stage_pairs[i]) (void)hipEventDestroy(c->ev_stage_pairs[i]);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int lf_ctx_set_params(lf_ctx *c, const lf_params *p) {
  if (!c || !p) return LF_ERR_INVALID;
  if (p->lsd_scale != c->params.lsd_scale || p->lsd_sigma_scale != c->params.lsd_sigma_scale)
    return LF_ERR_UNSUPPORTED;   // buffer geometry is fixed at creation
  lf_params old = c->params;
  // the device tables (Gaussian taps, log-gamma, nfa) depend on the lsd_* members only: a change of any other
  // parameter (Node::detect3DLines passes its scalars with every frame) costs nothing
  const bool lsd_same = p->lsd_angle_th == old.lsd_angle_th && p->lsd_density_th == old.lsd_density_th &&
                        p->lsd_quant == old.lsd_quant && p->lsd_log_eps == old.lsd_log_eps && p->lsd_n_bins == old.lsd_n_bins &&
                        p->lsd_max_grad == old.lsd_max_grad;
  c->params = *p;
  if (lsd_same) return LF_OK;
  int r = build_lsd_consts(c);
  if (r != LF_OK) { c->params = old; build_lsd_consts(c); return r; }
  HIPCHK(c, hipSetDevice(c->device));
  return upload_lsd_tables(c);
}

int lf_ctx_synchronize(lf_ctx *c) {
  if (!c) return LF_ERR_INVALID;
  pt_stream_join(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LF_OK;
}

int lf_lsd_batch_device(lf_ctx *c, const uint8_t *d_gray, size_t frame_stride, int row_stride,
                        int n_frames) {
  if (!c || !d_gray || n_frames < 1 || row_stride < c->W) return LF_ERR_INVALID;
  if (n_frames > c->maxB) return LF_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  c->lb.gray = d_gray;
  c->lb.gray_frame_stride = frame_stride;
  c->lb.gray_row_stride = row_stride;
  c->lb.ev_pre = c->ev[0]; c->lb.ev_sweep0 = c->ev[1]; c->lb.ev_sweep1 = c->ev[2];
  LsdConsts lc = c->lc;
  if (lc.sweep_waves == 0) lc.sweep_waves = (n_frames <= 160) ? LF_MW_MAXW : 1;
  lf_lsd_launch(lc, c->lb, n_frames, c->stream);
  HIPCHK(c, hipGetLastError());
  c->last_batch = n_frames;
  return LF_OK;
}

int lf_lsd_dims(const lf_ctx *c, int *N, int *M) {
  if (!c) return LF_ERR_INVALID;
  if (N) *N = c->lc.N;
  if (M) *M = c->lc.M;
  return LF_OK;
}

int lf_lsd_get_segments(lf_ctx *c, int frame, double *segs, int cap, int *n_out) {
  if (!c || frame < 0 || frame >= c->last_batch || !n_out || cap < 0 || (cap > 0 && !segs)) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->lb.nsegs + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = n;
  int m = n < cap ? n : cap;
  if (m > c->lc.seg_cap) m = c->lc.seg_cap;
  if (m > 0) {
    HIPCHK(c, hipMemcpyAsync(segs, c->lb.segs + (size_t)frame * c->lc.seg_cap * LF_SEG_STRIDE,
                             (size_t)m * LF_SEG_STRIDE * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return (n > cap || n > c->lc.seg_cap) ? LF_ERR_CAPACITY : LF_OK;
}

int lf_lsd_get_labels(lf_ctx *c, int frame, uint16_t *labels) {
  if (!c || frame < 0 || frame >= c->last_batch || !labels) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t NM = (size_t)c->lc.N * c->lc.M;
  HIPCHK(c, hipMemcpyAsync(labels, c->lb.labels + frame * NM, NM * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LF_OK;
}

int lf_lsd_get_debug(lf_ctx *c, int frame, int which, void *out, size_t out_bytes, int *count) {
  if (!c || frame < 0 || frame >= c->last_batch || !out) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t NM = (size_t)c->lc.N * c->lc.M;
  const void *src = nullptr;
  size_t bytes = 0;
  int cnt = 0;
  switch (which) {
    case 0: src = c->lb.scaled + frame * NM; bytes = NM * 8; cnt = (int)NM; break;
    case 1: src = c->lb.angles + frame * NM; bytes = NM * 8; cnt = (int)NM; break;
    case 2: src = c->lb.modgrad + frame * NM; bytes = NM * 8; cnt = (int)NM; break;
    case 3: {
      HIPCHK(c, hipMemcpyAsync(&cnt, c->lb.nseeds + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      src = c->lb.seeds + frame * NM; bytes = (size_t)cnt * 4; break;
    }
    case 4: src = c->lb.stats + (size_t)frame * LF_STATS_STRIDE; bytes = 8 * LF_STATS_STRIDE; cnt = LF_STATS_STRIDE; break;
    default: return LF_ERR_INVALID;
  }
  if (count) *count = cnt;
  if (out_bytes < bytes) return LF_ERR_CAPACITY;
  if (bytes) {
    HIPCHK(c, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LF_OK;
}

int lf_lsd(lf_ctx *c, const uint8_t *gray, int row_stride, int width, int height, double *segs, int cap,
           int *n_out, uint16_t *labels) {
  if (!c || !gray || width != c->W || height != c->H || row_stride < width) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy2DAsync(c->d_gray_stage, (size_t)width, gray, (size_t)row_stride, (size_t)width,
                             (size_t)height, hipMemcpyHostToDevice, c->stream));
  int r = lf_lsd_batch_device(c, c->d_gray_stage, (size_t)width * height, width, 1);
  if (r != LF_OK) return r;
  r = lf_lsd_get_segments(c, 0, segs, cap, n_out);
  if (r != LF_OK && r != LF_ERR_CAPACITY) return r;
  if (labels) {
    int r2 = lf_lsd_get_labels(c, 0, labels);
    if (r2 != LF_OK) return r2;
  }
  return r;
}

// ---- a9-a18 ------------------------------------------------------------------------------------
static void set_camera(lf_ctx *c, const double K[9]) {
  FrontConsts &fc = c->fc;
  for (int i = 0; i < 9; i++) fc.K[i] = K[i];
  // Eigen::Matrix3d::inverse() as used at lineslam.cpp:238-242: cofactors / determinant
  double c00 = K[4] * K[8] - K[5] * K[7], c10 = K[2] * K[7] - K[1] * K[8], c20 = K[1] * K[5] - K[2] * K[4];
  double det = c00 * K[0] + c10 * K[3] + c20 * K[6], inv = 1.0 / det;
  fc.Kinv[0] = c00 * inv; fc.Kinv[1] = c10 * inv; fc.Kinv[2] = c20 * inv;
  fc.Kinv[3] = (K[5] * K[6] - K[3] * K[8]) * inv; fc.Kinv[4] = (K[0] * K[8] - K[2] * K[6]) * inv; fc.Kinv[5] = (K[2] * K[3] - K[0] * K[5]) * inv;
  fc.Kinv[6] = (K[3] * K[7] - K[4] * K[6]) * inv; fc.Kinv[7] = (K[1] * K[6] - K[0] * K[7]) * inv; fc.Kinv[8] = (K[0] * K[4] - K[1] * K[3]) * inv;
  fc.P = c->params;
}

int lf_detect3d_batch_device(lf_ctx *c, const uint8_t *d_gray, size_t gray_frame_stride, int gray_row_stride,
                             const float *d_depth, size_t depth_frame_stride, int depth_row_stride,
                             int n_frames, const double K[9], const uint64_t *frame_ids) {
  if (!c || !d_gray || !d_depth || !K || n_frames < 1 || gray_row_stride < c->W || depth_row_stride < c->W)
    return LF_ERR_INVALID;
  if (n_frames > c->maxB) return LF_ERR_CAPACITY;
  if (c->params.line_sample_max_num + 1 > LF_MAX_SAMPLES) return LF_ERR_UNSUPPORTED;
  if (c->params.line_detector != LF_DETECTOR_LSD && c->params.line_detector != LF_DETECTOR_EDLINES) return LF_ERR_UNSUPPORTED;
  int r = c->params.line_detector == LF_DETECTOR_EDLINES ? lf_edlines_batch_device(c, d_gray, gray_frame_stride, gray_row_stride, n_frames)
                                                         : lf_lsd_batch_device(c, d_gray, gray_frame_stride, gray_row_stride, n_frames);
  if (r != LF_OK) return r;
  {   // node ids through the pinned staging area (no host synchronisation)
    if (c->stage_ids_pending) HIPCHK(c, hipEventSynchronize(c->ev_stage_ids));
    uint64_t *ids = (uint64_t *)c->h_stage;
    for (int i = 0; i < n_frames; i++) ids[i] = frame_ids ? frame_ids[i] : (uint64_t)i;
    HIPCHK(c, hipMemcpyAsync(c->d_frame_ids, ids, (size_t)n_frames * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_stage_ids, c->stream));
    c->stage_ids_pending = true;
  }
  set_camera(c, K);
  c->fb.gray = d_gray; c->fb.gray_frame_stride = gray_frame_stride; c->fb.gray_row_stride = gray_row_stride;
  c->fb.depth = d_depth; c->fb.depth_frame_stride = depth_frame_stride; c->fb.depth_row_stride = depth_row_stride;
  lf_front_launch(c->fc, c->fb, n_frames, c->stream);
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  HIPCHK(c, hipGetLastError());
  return LF_OK;
}

int lf_frame_get_lines(lf_ctx *c, int frame, lf_line_record *out, int cap, int *n_out) {
  if (!c || frame < 0 || frame >= c->last_batch || !n_out || cap < 0 || (cap > 0 && !out)) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int n = 0, nseg = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->fb.nlines + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&nseg, c->lb.nsegs + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = n;
  int m = n < cap ? n : cap;
  if (m > c->fc.line_cap) m = c->fc.line_cap;
  if (m > 0) {
    HIPCHK(c, hipMemcpyAsync(out, c->fb.recs + (size_t)frame * c->fc.line_cap, (size_t)m * sizeof(lf_line_record),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return (n > cap || n > c->fc.line_cap || nseg > c->fc.seg_cap) ? LF_ERR_CAPACITY : LF_OK;   // nothing is dropped silently
}

int lf_frame_get_candidates(lf_ctx *c, int frame, int32_t *flags, double *info, int cap, int *n_out) {
  if (!c || frame < 0 || frame >= c->last_batch || !n_out || cap < 0) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->lb.nsegs + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const bool seg_overflow = n > c->fc.cand_cap;      // LSD itself found more than seg_cap segments (lf_lsd_get_segments says so too)
  if (n > c->fc.cand_cap) n = c->fc.cand_cap;
  *n_out = n;
  int m = n < cap ? n : cap;
  if (m > 0 && flags) HIPCHK(c, hipMemcpyAsync(flags, c->fb.cand_flag + (size_t)frame * c->fc.cand_cap, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (m > 0 && info) HIPCHK(c, hipMemcpyAsync(info, c->fb.cand_out + (size_t)frame * c->fc.cand_cap * LF_CAND_STRIDE, (size_t)m * LF_CAND_STRIDE * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return (n > cap || seg_overflow) ? LF_ERR_CAPACITY : LF_OK;
}

int lf_detect3d(lf_ctx *c, const uint8_t *gray, int gray_row_stride, const float *depth_m, int depth_row_stride,
                int width, int height, const double K[9], uint64_t frame_id, lf_line_record *out, int cap,
                int *n_out) {
  if (!c || !gray || !depth_m || width != c->W || height != c->H || gray_row_stride < width || depth_row_stride < width)
    return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy2DAsync(c->d_gray_stage, (size_t)width, gray, (size_t)gray_row_stride, (size_t)width,
                             (size_t)height, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpy2DAsync(c->d_depth_stage, (size_t)width * 4, depth_m, (size_t)depth_row_stride * 4,
                             (size_t)width * 4, (size_t)height, hipMemcpyHostToDevice, c->stream));
  int r = lf_detect3d_batch_device(c, c->d_gray_stage, (size_t)width * height, width, c->d_depth_stage,
                                   (size_t)width * height, width, 1, K, &frame_id);
  if (r != LF_OK) return r;
  return lf_frame_get_lines(c, 0, out, cap, n_out);
}

// ---- a19-a25 -----------------------------------------------------------------------------------
struct HybridArgs { const float *d_points; int pt_cap; const int32_t *pm_q, *pm_t, *n_pm; int pm_cap; const double *K; bool pm_on_device; };
#define LF_NODE_PT_CAP 4096

static int hybrid_prepare(lf_ctx *c, const HybridArgs &h, int n_pairs, PairBuffers &pb) {
  if (!h.d_points || h.pt_cap < 1 || !h.pm_q || !h.pm_t || !h.n_pm || h.pm_cap < 0 || !h.K) return LF_ERR_INVALID;
  for (int i = 0; i < n_pairs && !h.pm_on_device; i++) {
    if (h.n_pm[i] < 0) return LF_ERR_INVALID;
    if (h.n_pm[i] > h.pm_cap) return LF_ERR_INVALID;
    if (h.n_pm[i] > c->pcn.pt_match_cap) { c->err = "point matches per pair exceed the context's pt_match_cap"; return LF_ERR_CAPACITY; }
    for (int k = 0; k < h.n_pm[i]; k++) {
      int a = h.pm_q[(size_t)i * h.pm_cap + k], b = h.pm_t[(size_t)i * h.pm_cap + k];
      if (a < 0 || a >= h.pt_cap || b < 0 || b >= h.pt_cap) { c->err = "point match index outside the point array"; return LF_ERR_INVALID; }
    }
  }
  if (!c->hybrid_ready) {
    const size_t B = (size_t)c->maxB;
    ALLOC(c, c->d_pm_q, B * LF_MAX_PT_MATCHES); ALLOC(c, c->d_pm_t, B * LF_MAX_PT_MATCHES); ALLOC(c, c->d_npm, B);
    ALLOC(c, c->pb.pt_inliers, B * LF_MAX_PT_MATCHES);
    ALLOC(c, c->pb.ws_h, B * lf_pair_hybrid_ws_doubles());
    ALLOC(c, c->d_pts_stage, (size_t)2 * LF_NODE_PT_CAP * 4);
    c->pb.pm_q = c->d_pm_q; c->pb.pm_t = c->d_pm_t; c->pb.npm = c->d_npm;
    c->hybrid_ready = true;
  }
  if (!h.pm_on_device) {
    std::vector<int> hq((size_t)n_pairs * LF_MAX_PT_MATCHES, 0), ht((size_t)n_pairs * LF_MAX_PT_MATCHES, 0);
    for (int i = 0; i < n_pairs; i++)
      for (int k = 0; k < h.n_pm[i]; k++) {
        hq[(size_t)i * LF_MAX_PT_MATCHES + k] = h.pm_q[(size_t)i * h.pm_cap + k];
        ht[(size_t)i * LF_MAX_PT_MATCHES + k] = h.pm_t[(size_t)i * h.pm_cap + k];
      }
    HIPCHK(c, hipMemcpyAsync(c->d_pm_q, hq.data(), hq.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pm_t, ht.data(), ht.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_npm, h.n_pm, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->pb.pm_q = c->d_pm_q; c->pb.pm_t = c->d_pm_t; c->pb.npm = c->d_npm; c->pb.pm_stride = LF_MAX_PT_MATCHES;
  pb = c->pb;
  if (h.pm_on_device) { pb.pm_q = h.pm_q; pb.pm_t = h.pm_t; pb.npm = h.n_pm; pb.pm_stride = h.pm_cap; }
  pb.pts = h.d_points; pb.pts_t = h.d_points; pb.pt_cap = h.pt_cap; pb.pt_cap_t = h.pt_cap;
  // errorFunction2 constants (misc.cpp:704-711) and sigma_depth (misc2.h:23), host libm as in the reference
  const double cam_angle_x = 58.0 / 180.0 * M_PI, cam_angle_y = 45.0 / 180.0 * M_PI;
  const double sx = 3 * tan(cam_angle_x / 640), sy = 3 * tan(cam_angle_y / 480);
  c->pcn.pm.raster_cov_x = sx * sx; c->pcn.pm.raster_cov_y = sy * sy; c->pcn.pm.sigma_depth = 0.01;
  c->pcn.focal = h.K[0];
  return LF_OK;
}

struct PairCall {
  const lf_line_record *d_ext_recs = nullptr; const int32_t *d_ext_nlines = nullptr; const uint64_t *d_ext_ids = nullptr;
  int ext_frames = 0, ext_line_cap = 0;                 // train side = an external device-resident map
  const HybridArgs *hy = nullptr;                       // point matches (hybrid solver)
  int solver = LF_SOLVER_LINES;
  const uint8_t *adjacent = nullptr;                    // HOST [n_pairs] adjacentFrame flags (null: from the node ids)
  const int32_t *lm_q = nullptr, *lm_t = nullptr, *n_lm = nullptr; int lm_cap = 0;   // HOST caller-supplied line matches
  int mode = LF_MODE_SOLVE, refine_iters = 0;
};

static int match_pairs_impl(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs, const PairCall &pcall) {
  if (!c || !query_frames || !train_frames || n_pairs < 1) return LF_ERR_INVALID;
  if (n_pairs > c->maxB) return LF_ERR_CAPACITY;
  const int ntrain = pcall.d_ext_recs ? pcall.ext_frames : c->last_batch;
  for (int i = 0; i < n_pairs; i++)
    if (query_frames[i] < 0 || query_frames[i] >= c->last_batch || train_frames[i] < 0 || train_frames[i] >= ntrain)
      return LF_ERR_INVALID;
  if (pcall.lm_q) {   // caller-supplied line matches: validated before anything is enqueued
    if (!pcall.lm_t || !pcall.n_lm || pcall.lm_cap < 0) return LF_ERR_INVALID;
    for (int i = 0; i < n_pairs; i++) {
      if (pcall.n_lm[i] < 0 || pcall.n_lm[i] > pcall.lm_cap) return LF_ERR_INVALID;
      if (pcall.n_lm[i] > c->pcn.match_cap) { c->err = "line matches per pair exceed the context's match_cap"; return LF_ERR_CAPACITY; }
      for (int k = 0; k < pcall.n_lm[i]; k++) {
        int a = pcall.lm_q[(size_t)i * pcall.lm_cap + k], b = pcall.lm_t[(size_t)i * pcall.lm_cap + k];
        if (a < 0 || a >= c->fc.line_cap || b < 0 || b >= (pcall.d_ext_recs ? pcall.ext_line_cap : c->fc.line_cap)) {
          c->err = "line match index outside the line map";
          return LF_ERR_INVALID;
        }
      }
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  if (pcall.hy) pt_stream_join(c);
  {   // pair lists through the pinned staging area (the caller's arrays may be temporaries)
    const unsigned sl = c->stage_pairs_next++ % LF_PAIR_STAGE_SLOTS;
    if (c->stage_pairs_pending[sl]) HIPCHK(c, hipEventSynchronize(c->ev_stage_pairs[sl]));
    int *hq = (int *)(c->h_stage + (size_t)c->maxB * 8 * (1 + sl)), *ht = hq + c->maxB;
    memcpy(hq, query_frames, sizeof(int) * (size_t)n_pairs);
    memcpy(ht, train_frames, sizeof(int) * (size_t)n_pairs);
    HIPCHK(c, hipMemcpyAsync(c->d_pair_q, hq, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pair_t, ht, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_stage_pairs[sl], c->stream));
    c->stage_pairs_pending[sl] = true;
  }
  c->pcn.P = c->params;
  c->pcn.mode = pcall.mode; c->pcn.refine_iters = pcall.refine_iters;
  PairBuffers pb = c->pb;
  if (pcall.hy) { int r = hybrid_prepare(c, *pcall.hy, n_pairs, pb); if (r != LF_OK) return r; }
  if (pcall.d_ext_recs) { pb.recs_t = pcall.d_ext_recs; pb.nlines_t = pcall.d_ext_nlines; pb.frame_ids_t = pcall.d_ext_ids; pb.line_cap_t = pcall.ext_line_cap; }
  pb.adjacent = nullptr;
  if (pcall.adjacent) {
    HIPCHK(c, hipMemcpyAsync(c->d_adjacent, pcall.adjacent, (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (the caller's array may be a temporary)
    pb.adjacent = c->d_adjacent;
  }
  if (pcall.lm_q) {
    std::vector<int> hq((size_t)n_pairs * c->pcn.match_cap, 0), ht((size_t)n_pairs * c->pcn.match_cap, 0);
    std::vector<double> hd((size_t)n_pairs * c->pcn.match_cap, 0.0);
    for (int i = 0; i < n_pairs; i++)
      for (int k = 0; k < pcall.n_lm[i]; k++) {
        hq[(size_t)i * c->pcn.match_cap + k] = pcall.lm_q[(size_t)i * pcall.lm_cap + k];
        ht[(size_t)i * c->pcn.match_cap + k] = pcall.lm_t[(size_t)i * pcall.lm_cap + k];
      }
    HIPCHK(c, hipMemcpyAsync(pb.match_q, hq.data(), hq.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(pb.match_t, ht.data(), ht.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(pb.match_d, hd.data(), hd.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(pb.nmatches, pcall.n_lm, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
  lf_pair_launch(c->pcn, pb, n_pairs, c->stream, pcall.solver, pcall.lm_q == nullptr);
  if (pcall.hy) pt_stream_consumed(c);
  c->last_hybrid = pcall.hy != nullptr;
  c->last_pb = pb;
  HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
  HIPCHK(c, hipGetLastError());
  c->last_pairs = n_pairs;
  c->pcn.mode = LF_MODE_SOLVE;
  return LF_OK;
}

int lf_match_pairs_device(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs) {
  return match_pairs_impl(c, query_frames, train_frames, n_pairs, PairCall());
}

int lf_match_external_device(lf_ctx *c, const int32_t *query_frames, const int32_t *train_slots, int n_pairs,
                             const lf_line_record *d_ext_recs, const int32_t *d_ext_nlines,
                             const uint64_t *d_ext_ids, int ext_frames, int ext_line_cap) {
  if (!d_ext_recs || !d_ext_nlines || !d_ext_ids || ext_frames < 1 || ext_line_cap < 1) return LF_ERR_INVALID;
  PairCall pc;
  pc.d_ext_recs = d_ext_recs; pc.d_ext_nlines = d_ext_nlines; pc.d_ext_ids = d_ext_ids; pc.ext_frames = ext_frames; pc.ext_line_cap = ext_line_cap;
  return match_pairs_impl(c, query_frames, train_slots, n_pairs, pc);
}

int lf_match_pairs_hybrid_device(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                                 const float *d_points, int pt_cap, const int32_t *pm_query, const int32_t *pm_train,
                                 const int32_t *n_pm, int pm_cap, const double K[9]) {
  if (!c) return LF_ERR_INVALID;
  HybridArgs h = {d_points, pt_cap, pm_query, pm_train, n_pm, pm_cap, K, false};
  PairCall pc;
  pc.hy = &h; pc.solver = LF_SOLVER_HYBRID;
  return match_pairs_impl(c, query_frames, train_frames, n_pairs, pc);
}

int lf_match_pairs_hybrid_device_pm(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                                    const float *d_points, int pt_cap, const int32_t *d_pm_query, const int32_t *d_pm_train,
                                    const int32_t *d_n_pm, int pm_stride, const double K[9]) {
  if (!c || pm_stride < 1) return LF_ERR_INVALID;
  HybridArgs h = {d_points, pt_cap, d_pm_query, d_pm_train, d_n_pm, pm_stride, K, true};
  PairCall pc;
  pc.hy = &h; pc.solver = LF_SOLVER_HYBRID;
  return match_pairs_impl(c, query_frames, train_frames, n_pairs, pc);
}

// ---- the point front end next to the line front end.  Node::Node extracts the key points while detect3DLines runs in a
// second thread (src/node.cpp:208-217, joined at :313-316).  Here: lf_orb_extract_device / lf_project_keypoints_device are
// enqueued on a second stream of the context; they start after everything enqueued on the context's stream BEFORE the call
// (their inputs), run beside what is enqueued after it (lf_detect3d_batch_device), and are joined -- stream-side, no host
// wait -- before their first consumer on the context's stream (feature matching, the hybrid solver, the getters) or by
// lf_ctx_point_join.  The next point-side call waits for the consumers of the previous one.
static hipStream_t pt_stream_begin(lf_ctx *c) {
  if (!c->pts_async) return c->stream;
  // EVERY point-side call starts after everything enqueued on the context's stream before it (the header's contract): a
  // second call before the join may read what was put on c->stream since the first one (an upload, another entry point)
  (void)hipEventRecord(c->ev_pts_in, c->stream);
  (void)hipStreamWaitEvent(c->pstream, c->ev_pts_in, 0);
  if (!c->pts_pending) {
    if (c->pts_free_rec) (void)hipStreamWaitEvent(c->pstream, c->ev_pts_free, 0);
    c->pts_pending = true;
  }
  return c->pstream;
}
static void pt_stream_join(lf_ctx *c) {
  if (!c->pts_async || !c->pts_pending) return;
  (void)hipEventRecord(c->ev_pts_done, c->pstream);
  (void)hipStreamWaitEvent(c->stream, c->ev_pts_done, 0);
  c->pts_pending = false;
}
static void pt_stream_consumed(lf_ctx *c) {      // the consumers of the point-side outputs are enqueued on c->stream
  if (!c->pts_async) return;
  (void)hipEventRecord(c->ev_pts_free, c->stream);
  c->pts_free_rec = true;
}
int lf_ctx_point_stream(lf_ctx *c, int enable) {
  if (!c) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  pt_stream_join(c);
  if (enable && !c->pstream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->pstream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pts_in, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pts_done, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pts_free, hipEventDisableTiming));
  }
  c->pts_async = enable != 0;
  return LF_OK;
}
int lf_ctx_point_join(lf_ctx *c) {
  if (!c) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  pt_stream_join(c);
  return LF_OK;
}

int lf_project_keypoints_device(lf_ctx *c, const float *d_depth, size_t depth_frame_stride, int depth_row_stride,
                                int n_frames, const float *d_kp_xy, const int32_t *d_nkp, int kp_cap, const double K[9],
                                double depth_scaling, int max_keypoints, float *d_points_out, int32_t *d_npts_out,
                                int32_t *d_kept_out) {
  if (!c || !d_depth || !d_kp_xy || !d_nkp || !K || !d_points_out || !d_npts_out || n_frames < 1 || kp_cap < 1 ||
      max_keypoints < 1 || depth_row_stride < c->W)
    return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  PointConsts pc;
  memset(&pc, 0, sizeof pc);
  pc.W = c->W; pc.H = c->H;
  for (int i = 0; i < 9; i++) pc.K[i] = K[i];
  pc.depth_scaling = depth_scaling; pc.max_keyp = max_keypoints < kp_cap ? max_keypoints : kp_cap; pc.kp_cap = kp_cap;
  PointBuffers pb;
  memset(&pb, 0, sizeof pb);
  pb.depth = d_depth; pb.depth_frame_stride = depth_frame_stride; pb.depth_row_stride = depth_row_stride;
  pb.kp_xy = d_kp_xy; pb.nkp = d_nkp; pb.points = d_points_out; pb.npts = d_npts_out; pb.kept = d_kept_out;
  lf_points_project_launch(pc, pb, n_frames, pt_stream_begin(c));
  HIPCHK(c, hipGetLastError());
  return LF_OK;
}

int lf_ingest_tum_device(lf_ctx *c, const uint8_t *d_rgb, const uint16_t *d_depth16, int n_frames, double depth_factor,
                         uint8_t *d_gray_out, float *d_depth_out) {
  if (!c || !d_rgb || !d_depth16 || !d_gray_out || !d_depth_out || n_frames < 1 || !(depth_factor > 0)) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  lf_points_ingest_launch(d_rgb, d_depth16, d_gray_out, d_depth_out, (size_t)n_frames * c->W * c->H, depth_factor, c->stream);
  HIPCHK(c, hipGetLastError());
  return LF_OK;
}

int lf_feature_match_pairs_device(lf_ctx *c, const uint8_t *d_desc, const int32_t *d_ndesc, int desc_cap,
                                  const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                                  double nn_distance_ratio, int32_t *d_match_q, int32_t *d_match_t, float *d_match_dist,
                                  int32_t *d_nmatch) {
  if (!c || !d_desc || !d_ndesc || !query_frames || !train_frames || !d_match_q || !d_match_t || !d_match_dist ||
      !d_nmatch || n_pairs < 1 || desc_cap < 1)
    return LF_ERR_INVALID;
  if (desc_cap > 1024 || n_pairs > c->maxB) return LF_ERR_CAPACITY;
  for (int i = 0; i < n_pairs; i++)
    if (query_frames[i] < 0 || query_frames[i] >= c->last_batch || train_frames[i] < 0 || train_frames[i] >= c->last_batch)
      return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  pt_stream_join(c);
  {   // pair lists through the pinned staging area
    const unsigned sl = c->stage_pairs_next++ % LF_PAIR_STAGE_SLOTS;
    if (c->stage_pairs_pending[sl]) HIPCHK(c, hipEventSynchronize(c->ev_stage_pairs[sl]));
    int *hq = (int *)(c->h_stage + (size_t)c->maxB * 8 * (1 + sl)), *ht = hq + c->maxB;
    memcpy(hq, query_frames, sizeof(int) * (size_t)n_pairs);
    memcpy(ht, train_frames, sizeof(int) * (size_t)n_pairs);
    HIPCHK(c, hipMemcpyAsync(c->d_pair_q, hq, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pair_t, ht, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_stage_pairs[sl], c->stream));
    c->stage_pairs_pending[sl] = true;
  }
  PointConsts pc;
  memset(&pc, 0, sizeof pc);
  pc.desc_cap = desc_cap; pc.nn_ratio = nn_distance_ratio; pc.rng_seed = c->params.rng_seed;
  PointBuffers pb;
  memset(&pb, 0, sizeof pb);
  pb.desc = d_desc; pb.ndesc = d_ndesc; pb.frame_ids = c->d_frame_ids; pb.pair_q = c->d_pair_q; pb.pair_t = c->d_pair_t;
  pb.fm_q = d_match_q; pb.fm_t = d_match_t; pb.fm_d = d_match_dist; pb.fm_n = d_nmatch;
  lf_points_match_launch(pc, pb, n_pairs, c->stream);
  HIPCHK(c, hipGetLastError());
  return LF_OK;
}

// Node::featureMatching for two HOST-resident nodes (descriptors of the newer node = query, of the older = train): the
// descriptors travel to a staging area laid out as frame slots 0 / 1, the node ids key the random distance offset.
int lf_feature_match_node_pair(lf_ctx *c, const uint8_t *desc_newer, int n_newer, uint64_t id_newer, const uint8_t *desc_older,
                               int n_older, uint64_t id_older, double nn_distance_ratio, int32_t *query_idx, int32_t *train_idx,
                               float *dist, int cap, int *n_out) {
  if (!c || !n_out || n_newer < 0 || n_older < 0 || (n_newer && !desc_newer) || (n_older && !desc_older) || cap < 0 ||
      (cap && (!query_idx || !train_idx || !dist)))
    return LF_ERR_INVALID;
  const int DC = 1024;                                   // desc_cap of the staging area = the matcher's maximum
  if (n_newer > DC || n_older > DC) return LF_ERR_CAPACITY;
  if (c->maxB < 2) return LF_ERR_CAPACITY;
  *n_out = 0;
  if (n_newer == 0 || n_older == 0) return LF_OK;        // (the reference returns with no matches: node.cpp:573-577)
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_fm_stage) {
    ALLOC(c, c->d_fm_stage, (size_t)2 * DC * 32);
    ALLOC(c, c->d_fm_n, 2);
    ALLOC(c, c->d_fm_q, DC); ALLOC(c, c->d_fm_t, DC); ALLOC(c, c->d_fm_d, DC); ALLOC(c, c->d_fm_cnt, 1);
  }
  const uint64_t ids[2] = {id_newer, id_older};
  const int32_t nd[2] = {n_newer, n_older};
  HIPCHK(c, hipMemcpyAsync(c->d_fm_stage, desc_newer, (size_t)32 * n_newer, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_fm_stage + (size_t)DC * 32, desc_older, (size_t)32 * n_older, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_fm_n, nd, sizeof nd, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_frame_ids, ids, sizeof ids, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->last_batch < 2) c->last_batch = 2;
  const int32_t q = 0, t = 1;
  int r = lf_feature_match_pairs_device(c, c->d_fm_stage, c->d_fm_n, DC, &q, &t, 1, nn_distance_ratio, c->d_fm_q, c->d_fm_t, c->d_fm_d, c->d_fm_cnt);
  if (r != LF_OK) return r;
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->d_fm_cnt, sizeof n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = n;
  if (n > cap) return LF_ERR_CAPACITY;
  if (n) {
    HIPCHK(c, hipMemcpyAsync(query_idx, c->d_fm_q, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(train_idx, c->d_fm_t, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dist, c->d_fm_d, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LF_OK;
}

int lf_relmotion_pairs_device(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs) {
  PairCall pc;
  pc.solver = LF_SOLVER_RELMOTION;
  return match_pairs_impl(c, query_frames, train_frames, n_pairs, pc);
}

// computeRelativeMotion_Ransac(a, b, Ro, to) for two HOST vectors of already matched lines (a[i] <-> b[i]).
int lf_relmotion_lines(lf_ctx *c, const lf_line_record *a, const lf_line_record *b, int n, uint64_t id_a, uint64_t id_b,
                       double R[9], double t[3], int32_t *inliers, int cap, int *n_inliers) {
  if (!c || !R || !t || !n_inliers || n < 0 || (n && (!a || !b)) || cap < 0) return LF_ERR_INVALID;
  if (c->maxB < 2) return LF_ERR_CAPACITY;
  if (n > c->fc.line_cap || n > LF_MAX_MATCHES || n > c->pcn.match_cap) return LF_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t ids[2] = {id_a, id_b};
  const int nl[2] = {n, n}, pq = 0, pt = 1;
  std::vector<int> iota((size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) iota[i] = i;
  if (n) {
    HIPCHK(c, hipMemcpyAsync(c->fb.recs, a, sizeof(lf_line_record) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->fb.recs + c->fc.line_cap, b, sizeof(lf_line_record) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pb.match_q, iota.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pb.match_t, iota.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipMemcpyAsync(c->fb.nlines, nl, sizeof nl, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_frame_ids, ids, sizeof ids, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->pb.nmatches, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_pair_q, &pq, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_pair_t, &pt, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->last_batch < 2) c->last_batch = 2;
  c->pcn.P = c->params;
  lf_pair_relmotion_launch(c->pcn, c->pb, 1, c->stream);
  HIPCHK(c, hipGetLastError());
  c->last_pairs = 1;
  c->last_hybrid = false;
  c->last_pb = c->pb;
  lf_pair_result r;
  int rc = lf_pair_get_result(c, 0, &r);
  if (rc != LF_OK) return rc;
  *n_inliers = r.n_inliers;
  if (r.n_inliers > 0) {
    rc = lf_pair_get_motion(c, 0, R, t);
    if (rc != LF_OK) return rc;
    int m = 0;
    return lf_pair_get_inliers(c, 0, inliers, cap, &m);
  }
  return LF_OK;
}

int lf_pair_get_motion(lf_ctx *c, int pair, double R[9], double t[3]) {
  if (!c || !R || !t || pair < 0 || pair >= c->last_pairs) return LF_ERR_INVALID;
  double h[LF_MOTION_STRIDE];
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(h, c->pb.motion_d + (size_t)pair * LF_MOTION_STRIDE, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 9; i++) R[i] = h[i];
  for (int i = 0; i < 3; i++) t[i] = h[9 + i];
  return LF_OK;
}

int lf_pair_get_point_inliers(lf_ctx *c, int pair, int32_t *match_idx, int cap, int *n_out) {
  if (!c || !n_out || pair < 0 || pair >= c->last_pairs || cap < 0) return LF_ERR_INVALID;
  lf_pair_result r;
  int rc = lf_pair_get_result(c, pair, &r);
  if (rc != LF_OK && rc != LF_ERR_CAPACITY) return rc;
  *n_out = r.n_point_inliers;
  int m = r.n_point_inliers < cap ? r.n_point_inliers : cap;
  if (m > 0 && match_idx && c->last_hybrid) {
    HIPCHK(c, hipMemcpyAsync(match_idx, c->pb.pt_inliers + (size_t)pair * LF_MAX_PT_MATCHES, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return (r.n_point_inliers > cap || r.overflow) ? LF_ERR_CAPACITY : LF_OK;
}

int lf_get_device_records(lf_ctx *c, lf_line_record **d_recs, int32_t **d_nlines, uint64_t **d_ids, int *line_cap) {
  if (!c) return LF_ERR_INVALID;
  if (d_recs) *d_recs = c->fb.recs;
  if (d_nlines) *d_nlines = c->fb.nlines;
  if (d_ids) *d_ids = c->d_frame_ids;
  if (line_cap) *line_cap = c->fc.line_cap;
  return LF_OK;
}

int lf_pair_get_result(lf_ctx *c, int pair, lf_pair_result *out) {
  if (!c || !out || pair < 0 || pair >= c->last_pairs) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(out, c->pb.results + pair, sizeof(lf_pair_result), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return out->overflow ? LF_ERR_CAPACITY : LF_OK;   // the record is complete either way; `overflow` says what was cut
}

int lf_pair_get_matches(lf_ctx *c, int pair, int32_t *qi, int32_t *ti, double *dist, int cap, int *n_out) {
  if (!c || !n_out || pair < 0 || pair >= c->last_pairs || cap < 0) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->pb.nmatches + pair, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_out = n;
  int m = n < cap ? n : cap;
  if (m > c->pcn.match_cap) m = c->pcn.match_cap;
  size_t off = (size_t)pair * c->pcn.match_cap;
  if (m > 0 && qi) HIPCHK(c, hipMemcpyAsync(qi, c->pb.match_q + off, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
  if (m > 0 && ti) HIPCHK(c, hipMemcpyAsync(ti, c->pb.match_t + off, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
  if (m > 0 && dist) HIPCHK(c, hipMemcpyAsync(dist, c->pb.match_d + off, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return (n > cap || n > c->pcn.match_cap) ? LF_ERR_CAPACITY : LF_OK;
}

int lf_pair_get_inliers(lf_ctx *c, int pair, int32_t *match_idx, int cap, int *n_out) {
  if (!c || !n_out || pair < 0 || pair >= c->last_pairs || cap < 0) return LF_ERR_INVALID;
  lf_pair_result r;
  int rc = lf_pair_get_result(c, pair, &r);
  if (rc != LF_OK && rc != LF_ERR_CAPACITY) return rc;
  *n_out = r.n_inliers;
  int m = r.n_inliers < cap ? r.n_inliers : cap;
  if (m > 0 && match_idx) {
    HIPCHK(c, hipMemcpyAsync(match_idx, c->pb.inliers + (size_t)pair * LF_MAX_MATCHES, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return (r.n_inliers > cap || r.overflow) ? LF_ERR_CAPACITY : LF_OK;
}

int lf_pair_get_descdiff(lf_ctx *c, int pair, double *D, size_t cap_doubles, int *n_query, int *n_train) {
  if (!c || pair < 0 || pair >= c->last_pairs) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const PairBuffers &pb = c->last_pb;
  int pq = 0, pt = 0, n1 = 0, n2 = 0;
  HIPCHK(c, hipMemcpyAsync(&pq, c->d_pair_q + pair, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&pt, c->d_pair_t + pair, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyAsync(&n1, pb.nlines + pq, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&n2, pb.nlines_t + pt, sizeof(int), hipMemcpyDeviceToHost, c->stream));   // (train side: possibly an external map)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n1 > c->fc.line_cap) n1 = c->fc.line_cap;
  if (n2 > pb.line_cap_t) n2 = pb.line_cap_t;
  if (n2 > c->fc.line_cap) n2 = c->fc.line_cap;
  if (n_query) *n_query = n1;
  if (n_train) *n_train = n2;
  size_t need = (size_t)n1 * n2;
  if (!D) return LF_OK;
  if (cap_doubles < need) return LF_ERR_CAPACITY;
  if (need) {
    if (!c->d_descdiff) ALLOC(c, c->d_descdiff, (size_t)c->fc.line_cap * c->fc.line_cap);
    c->pcn.P = c->params;
    lf_pair_descdiff_launch(c->pcn, pb, pair, c->d_descdiff, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(D, c->d_descdiff, need * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LF_OK;
}

// Stage durations of the last launches, from HIP events recorded on the context stream:
// which = 0 LSD data-parallel kernels (sampler, gradient, seed sort), 1 k_lsd_sweep, 2 3D-line stage,
// 3 pair solver.  Synchronises the stream.
int lf_get_stage_ms(lf_ctx *c, int which, float *ms) {
  if (!c || !ms || which < 0 || which > 3) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  hipEvent_t a, b;
  switch (which) {
    case 0: a = c->ev[0]; b = c->ev[1]; break;
    case 1: a = c->ev[1]; b = c->ev[2]; break;
    case 2: a = c->ev[2]; b = c->ev[3]; break;
    default: a = c->ev[4]; b = c->ev[5]; break;
  }
  hipError_t e = hipEventElapsedTime(ms, a, b);
  if (e != hipSuccess) { *ms = -1.0f; return fail_hip(c, e, "hipEventElapsedTime"); }
  return LF_OK;
}

// Node-level convenience: two frames whose line maps live in HOST memory (e.g. two lf::Node objects).
int lf_match_node_pair(lf_ctx *c, const lf_line_record *newer, int n_newer, uint64_t id_newer,
                       const lf_line_record *older, int n_older, uint64_t id_older, lf_pair_result *out) {
  if (!c || !out || n_newer < 0 || n_older < 0 || (n_newer && !newer) || (n_older && !older)) return LF_ERR_INVALID;
  if (c->maxB < 2) return LF_ERR_CAPACITY;
  if (n_newer > c->fc.line_cap || n_older > c->fc.line_cap) return LF_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t ids[2] = {id_newer, id_older};
  const int nl[2] = {n_newer, n_older};
  if (n_newer) HIPCHK(c, hipMemcpyAsync(c->fb.recs, newer, sizeof(lf_line_record) * (size_t)n_newer, hipMemcpyHostToDevice, c->stream));
  if (n_older) HIPCHK(c, hipMemcpyAsync(c->fb.recs + c->fc.line_cap, older, sizeof(lf_line_record) * (size_t)n_older, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.nlines, nl, sizeof nl, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_frame_ids, ids, sizeof ids, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->last_batch < 2) c->last_batch = 2;
  const int32_t q = 0, t = 1;
  int r = lf_match_pairs_device(c, &q, &t, 1);
  if (r != LF_OK) return r;
  return lf_pair_get_result(c, 0, out);
}

int lf_match_node_pair_hybrid(lf_ctx *c, const lf_line_record *newer, int n_newer, uint64_t id_newer,
                              const float *pts_newer, int n_pts_newer, const lf_line_record *older, int n_older,
                              uint64_t id_older, const float *pts_older, int n_pts_older, const int32_t *pm_query,
                              const int32_t *pm_train, int n_pm, const double K[9], lf_pair_result *out) {
  if (!c || !out || n_newer < 0 || n_older < 0 || (n_newer && !newer) || (n_older && !older) || n_pts_newer < 0 ||
      n_pts_older < 0 || (n_pts_newer && !pts_newer) || (n_pts_older && !pts_older) || n_pm < 0 ||
      (n_pm && (!pm_query || !pm_train)) || !K)
    return LF_ERR_INVALID;
  if (c->maxB < 2) return LF_ERR_CAPACITY;
  if (n_newer > c->fc.line_cap || n_older > c->fc.line_cap || n_pts_newer > LF_NODE_PT_CAP || n_pts_older > LF_NODE_PT_CAP)
    return LF_ERR_CAPACITY;
  for (int k = 0; k < n_pm; k++)
    if (pm_query[k] < 0 || pm_query[k] >= n_pts_newer || pm_train[k] < 0 || pm_train[k] >= n_pts_older) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t ids[2] = {id_newer, id_older};
  const int nl[2] = {n_newer, n_older};
  if (n_newer) HIPCHK(c, hipMemcpyAsync(c->fb.recs, newer, sizeof(lf_line_record) * (size_t)n_newer, hipMemcpyHostToDevice, c->stream));
  if (n_older) HIPCHK(c, hipMemcpyAsync(c->fb.recs + c->fc.line_cap, older, sizeof(lf_line_record) * (size_t)n_older, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.nlines, nl, sizeof nl, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_frame_ids, ids, sizeof ids, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->last_batch < 2) c->last_batch = 2;
  // first hybrid call allocates the staging buffer: run an empty prepare through a dummy call path
  const int32_t q = 0, t = 1, npm = n_pm, zero = 0;
  if (!c->hybrid_ready) {
    PairBuffers tmp;
    float dummy_pt = 0;
    HybridArgs h0 = {&dummy_pt, 1, &zero, &zero, &zero, 0, K, false};
    int r0 = hybrid_prepare(c, h0, 1, tmp);
    if (r0 != LF_OK) return r0;
  }
  if (n_pts_newer) HIPCHK(c, hipMemcpyAsync(c->d_pts_stage, pts_newer, sizeof(float) * 4 * (size_t)n_pts_newer, hipMemcpyHostToDevice, c->stream));
  if (n_pts_older) HIPCHK(c, hipMemcpyAsync(c->d_pts_stage + (size_t)LF_NODE_PT_CAP * 4, pts_older, sizeof(float) * 4 * (size_t)n_pts_older, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int32_t *pq = n_pm ? pm_query : &zero, *pt = n_pm ? pm_train : &zero;
  int r = lf_match_pairs_hybrid_device(c, &q, &t, 1, c->d_pts_stage, LF_NODE_PT_CAP, pq, pt, &npm, n_pm > 0 ? n_pm : 1, K);
  if (r != LF_OK) return r;
  return lf_pair_get_result(c, 0, out);
}


// ---- the operators of the pair path on their own ---------------------------------------------------------------------
int lf_line_matching_device(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                            const uint8_t *adjacent, const lf_line_record *d_ext_recs, const int32_t *d_ext_nlines,
                            const uint64_t *d_ext_ids, int ext_frames, int ext_line_cap) {
  if (!c) return LF_ERR_INVALID;
  PairCall pc;
  if (d_ext_recs) {
    if (!d_ext_nlines || !d_ext_ids || ext_frames < 1 || ext_line_cap < 1) return LF_ERR_INVALID;
    pc.d_ext_recs = d_ext_recs; pc.d_ext_nlines = d_ext_nlines; pc.d_ext_ids = d_ext_ids; pc.ext_frames = ext_frames; pc.ext_line_cap = ext_line_cap;
  }
  pc.solver = LF_SOLVER_NONE;
  pc.adjacent = adjacent;
  return match_pairs_impl(c, query_frames, train_frames, n_pairs, pc);
}

int lf_solve_pairs_device(lf_ctx *c, const int32_t *query_frames, const int32_t *train_frames, int n_pairs,
                          const int32_t *lm_query, const int32_t *lm_train, const int32_t *n_lm, int lm_cap,
                          const float *d_points, int pt_cap, const int32_t *pm_query, const int32_t *pm_train,
                          const int32_t *n_pm, int pm_cap, const double K[9]) {
  if (!c || !lm_query || !lm_train || !n_lm || lm_cap < 0) return LF_ERR_INVALID;
  PairCall pc;
  pc.lm_q = lm_query; pc.lm_t = lm_train; pc.n_lm = n_lm; pc.lm_cap = lm_cap;
  if (d_points) {
    HybridArgs h = {d_points, pt_cap, pm_query, pm_train, n_pm, pm_cap, K, false};
    pc.hy = &h; pc.solver = LF_SOLVER_HYBRID;
    return match_pairs_impl(c, query_frames, train_frames, n_pairs, pc);
  }
  return match_pairs_impl(c, query_frames, train_frames, n_pairs, pc);
}

// both nodes of a pair from HOST memory into frame slots 0 (newer) / 1 (older) (+ their 3D points into the staging area)
static int upload_node_pair(lf_ctx *c, const lf_line_record *newer, int n_newer, uint64_t id_newer, const float *pts_newer,
                            int n_pts_newer, const lf_line_record *older, int n_older, uint64_t id_older,
                            const float *pts_older, int n_pts_older, const double K[9], bool want_points) {
  if (n_newer < 0 || n_older < 0 || (n_newer && !newer) || (n_older && !older) || n_pts_newer < 0 || n_pts_older < 0 ||
      (n_pts_newer && !pts_newer) || (n_pts_older && !pts_older))
    return LF_ERR_INVALID;
  if (c->maxB < 2) return LF_ERR_CAPACITY;
  if (n_newer > c->fc.line_cap || n_older > c->fc.line_cap || n_pts_newer > LF_NODE_PT_CAP || n_pts_older > LF_NODE_PT_CAP)
    return LF_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  const uint64_t ids[2] = {id_newer, id_older};
  const int nl[2] = {n_newer, n_older};
  if (n_newer) HIPCHK(c, hipMemcpyAsync(c->fb.recs, newer, sizeof(lf_line_record) * (size_t)n_newer, hipMemcpyHostToDevice, c->stream));
  if (n_older) HIPCHK(c, hipMemcpyAsync(c->fb.recs + c->fc.line_cap, older, sizeof(lf_line_record) * (size_t)n_older, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.nlines, nl, sizeof nl, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_frame_ids, ids, sizeof ids, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->last_batch < 2) c->last_batch = 2;
  if (want_points) {
    if (!c->hybrid_ready) {   // first hybrid call allocates the staging buffers
      PairBuffers tmp;
      float dummy_pt = 0;
      const int32_t zero = 0;
      HybridArgs h0 = {&dummy_pt, 1, &zero, &zero, &zero, 0, K, false};
      int r0 = hybrid_prepare(c, h0, 1, tmp);
      if (r0 != LF_OK) return r0;
    }
    if (n_pts_newer) HIPCHK(c, hipMemcpyAsync(c->d_pts_stage, pts_newer, sizeof(float) * 4 * (size_t)n_pts_newer, hipMemcpyHostToDevice, c->stream));
    if (n_pts_older) HIPCHK(c, hipMemcpyAsync(c->d_pts_stage + (size_t)LF_NODE_PT_CAP * 4, pts_older, sizeof(float) * 4 * (size_t)n_pts_older, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LF_OK;
}

int lf_line_matching_node_pair(lf_ctx *c, const lf_line_record *query, int n_query, uint64_t id_query,
                               const lf_line_record *train, int n_train, uint64_t id_train, int adjacent, int32_t *query_idx,
                               int32_t *train_idx, double *dist, int cap, int *n_out) {
  if (!c || !n_out || cap < 0 || adjacent < -1 || adjacent > 1) return LF_ERR_INVALID;
  const double K0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  int r = upload_node_pair(c, query, n_query, id_query, nullptr, 0, train, n_train, id_train, nullptr, 0, K0, false);
  if (r != LF_OK) return r;
  const int32_t q = 0, t = 1;
  const uint8_t adj = (uint8_t)adjacent;
  r = lf_line_matching_device(c, &q, &t, 1, adjacent < 0 ? nullptr : &adj, nullptr, nullptr, nullptr, 0, 0);
  if (r != LF_OK) return r;
  return lf_pair_get_matches(c, 0, query_idx, train_idx, dist, cap, n_out);
}

int lf_solve_node_pair(lf_ctx *c, const lf_line_record *newer, int n_newer, uint64_t id_newer, const float *pts_newer,
                       int n_pts_newer, const lf_line_record *older, int n_older, uint64_t id_older, const float *pts_older,
                       int n_pts_older, const int32_t *lm_query, const int32_t *lm_train, int n_lm, const int32_t *pm_query,
                       const int32_t *pm_train, int n_pm, const double K[9], lf_pair_result *out) {
  if (!c || !out || !K || n_lm < 0 || n_pm < 0 || (n_lm && (!lm_query || !lm_train)) || (n_pm && (!pm_query || !pm_train)))
    return LF_ERR_INVALID;
  for (int k = 0; k < n_lm; k++) if (lm_query[k] < 0 || lm_query[k] >= n_newer || lm_train[k] < 0 || lm_train[k] >= n_older) return LF_ERR_INVALID;
  for (int k = 0; k < n_pm; k++) if (pm_query[k] < 0 || pm_query[k] >= n_pts_newer || pm_train[k] < 0 || pm_train[k] >= n_pts_older) return LF_ERR_INVALID;
  const bool pts = n_pm > 0;
  int r = upload_node_pair(c, newer, n_newer, id_newer, pts_newer, n_pts_newer, older, n_older, id_older, pts_older, n_pts_older, K, pts);
  if (r != LF_OK) return r;
  const int32_t q = 0, t = 1, nl = n_lm, np = n_pm, zero = 0;
  r = lf_solve_pairs_device(c, &q, &t, 1, n_lm ? lm_query : &zero, n_lm ? lm_train : &zero, &nl, n_lm > 0 ? n_lm : 1,
                            pts ? c->d_pts_stage : nullptr, LF_NODE_PT_CAP, pm_query, pm_train, &np, n_pm > 0 ? n_pm : 1, K);
  if (r != LF_OK) return r;
  return lf_pair_get_result(c, 0, out);
}

int lf_refine_pair(lf_ctx *c, const lf_line_record *newer, int n_newer, const float *pts_newer, int n_pts_newer,
                   const lf_line_record *older, int n_older, const float *pts_older, int n_pts_older,
                   const int32_t *lm_query, const int32_t *lm_train, int n_lm, const int32_t *pm_query, const int32_t *pm_train,
                   int n_pm, const double K[9], float T[16], int iterations) {
  if (!c || !T || !K || iterations < 0 || n_lm < 0 || n_pm < 0 || (n_lm && (!lm_query || !lm_train)) || (n_pm && (!pm_query || !pm_train)))
    return LF_ERR_INVALID;
  for (int k = 0; k < n_lm; k++) if (lm_query[k] < 0 || lm_query[k] >= n_newer || lm_train[k] < 0 || lm_train[k] >= n_older) return LF_ERR_INVALID;
  for (int k = 0; k < n_pm; k++) if (pm_query[k] < 0 || pm_query[k] >= n_pts_newer || pm_train[k] < 0 || pm_train[k] >= n_pts_older) return LF_ERR_INVALID;
  int r = upload_node_pair(c, newer, n_newer, 1, pts_newer, n_pts_newer, older, n_older, 0, pts_older, n_pts_older, K, true);
  if (r != LF_OK) return r;
  lf_pair_result seed;
  memset(&seed, 0, sizeof seed);
  for (int i = 0; i < 16; i++) seed.T[i] = T[i];
  HIPCHK(c, hipMemcpyAsync(c->pb.results, &seed, sizeof seed, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int32_t q = 0, t = 1, nl = n_lm, np = n_pm, zero = 0;
  HybridArgs h = {c->d_pts_stage, LF_NODE_PT_CAP, n_pm ? pm_query : &zero, n_pm ? pm_train : &zero, &np, n_pm > 0 ? n_pm : 1, K, false};
  PairCall pc;
  pc.lm_q = n_lm ? lm_query : &zero; pc.lm_t = n_lm ? lm_train : &zero; pc.n_lm = &nl; pc.lm_cap = n_lm > 0 ? n_lm : 1;
  pc.hy = &h; pc.solver = LF_SOLVER_HYBRID; pc.mode = LF_MODE_REFINE; pc.refine_iters = iterations;
  r = match_pairs_impl(c, &q, &t, 1, pc);
  if (r != LF_OK) return r;
  lf_pair_result res;
  r = lf_pair_get_result(c, 0, &res);
  if (r != LF_OK) return r;
  for (int i = 0; i < 16; i++) T[i] = res.T[i];
  return LF_OK;
}

int lf_relative_transformation_legacy(lf_ctx *c, const float *pts_newer, int n_pts_newer, uint64_t id_newer, const float *pts_older,
                                      int n_pts_older, uint64_t id_older, const int32_t *match_query, const int32_t *match_train,
                                      const float *match_dist, int n_matches, int min_matches, int ransac_iterations,
                                      double max_dist_for_inliers, int g2o_refinement_iterations, float T[16], float *rmse,
                                      int32_t *inlier_idx, int cap, int *n_inliers, int *found) {
  if (!c || !T || !rmse || !n_inliers || !found || n_matches < 0 || n_pts_newer < 0 || n_pts_older < 0 || ransac_iterations < 0 ||
      (n_matches && (!match_query || !match_train || !match_dist || !pts_newer || !pts_older)))
    return LF_ERR_INVALID;
  if (g2o_refinement_iterations > 0) {
    c->err = "the g2o step of getRelativeTransformationTo (EdgeSE3PointXYZDepth, node.cpp:1283-1327) is not restated: pass 0";
    return LF_ERR_UNSUPPORTED;
  }
  if (n_matches > LF_LEGACY_CAP || n_pts_newer > LF_NODE_PT_CAP || n_pts_older > LF_NODE_PT_CAP) {
    char m[160];
    snprintf(m, sizeof m, "legacy point RANSAC: %d matches / %d + %d points exceed the compiled capacities (%d matches, %d points per node)",
             n_matches, n_pts_newer, n_pts_older, LF_LEGACY_CAP, LF_NODE_PT_CAP);
    c->err = m;
    return LF_ERR_CAPACITY;
  }
  for (int k = 0; k < n_matches; k++) {
    if (match_query[k] < 0 || match_query[k] >= n_pts_newer || match_train[k] < 0 || match_train[k] >= n_pts_older) return LF_ERR_INVALID;
    // a NaN distance has no rank: the kernel's rank sort (std::sort of the reference is undefined there too) would leave
    // permutation slots unwritten and read points through them
    if (!(match_dist[k] == match_dist[k]) || match_dist[k] > 3.0e38f || match_dist[k] < -3.0e38f) {
      c->err = "legacy point RANSAC: a match distance is not finite";
      return LF_ERR_INVALID;
    }
  }
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_pts_stage) ALLOC(c, c->d_pts_stage, (size_t)2 * LF_NODE_PT_CAP * 4);
  // (each buffer on its own test: a failure of a later allocation must not leave the earlier pointer as the "all allocated" flag)
  if (!c->d_legacy_i) ALLOC(c, c->d_legacy_i, (size_t)3 * LF_LEGACY_CAP);
  if (!c->d_legacy_f) ALLOC(c, c->d_legacy_f, (size_t)LF_LEGACY_CAP);
  if (!c->d_legacy_r) ALLOC(c, c->d_legacy_r, 1);
  if (n_pts_newer) HIPCHK(c, hipMemcpyAsync(c->d_pts_stage, pts_newer, sizeof(float) * 4 * (size_t)n_pts_newer, hipMemcpyHostToDevice, c->stream));
  if (n_pts_older) HIPCHK(c, hipMemcpyAsync(c->d_pts_stage + (size_t)LF_NODE_PT_CAP * 4, pts_older, sizeof(float) * 4 * (size_t)n_pts_older, hipMemcpyHostToDevice, c->stream));
  if (n_matches) {
    HIPCHK(c, hipMemcpyAsync(c->d_legacy_i, match_query, sizeof(int) * (size_t)n_matches, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_legacy_i + LF_LEGACY_CAP, match_train, sizeof(int) * (size_t)n_matches, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_legacy_f, match_dist, sizeof(float) * (size_t)n_matches, hipMemcpyHostToDevice, c->stream));
  }
  LegacyArgs a;
  a.pts_q = c->d_pts_stage; a.pts_t = c->d_pts_stage + (size_t)LF_NODE_PT_CAP * 4;
  a.mq = c->d_legacy_i; a.mt = c->d_legacy_i + LF_LEGACY_CAP; a.md = c->d_legacy_f; a.n = n_matches;
  a.min_matches = min_matches; a.iterations = ransac_iterations; a.max_dist_m = (float)max_dist_for_inliers;
  a.seed = c->params.rng_seed; a.stream = ((id_newer << 32) ^ (uint64_t)(uint32_t)id_older ^ 0x5000000000000000ULL);
  {   // errorFunction2 constants (misc.cpp:704-711) and sigma_depth (misc2.h:23), host libm as in the reference
    const double cam_angle_x = 58.0 / 180.0 * M_PI, cam_angle_y = 45.0 / 180.0 * M_PI;
    const double sx = 3 * tan(cam_angle_x / 640), sy = 3 * tan(cam_angle_y / 480);
    a.pm.raster_cov_x = sx * sx; a.pm.raster_cov_y = sy * sy; a.pm.sigma_depth = 0.01;
  }
  a.out = c->d_legacy_r; a.out_inliers = c->d_legacy_i + 2 * LF_LEGACY_CAP;
  lf_legacy_launch(a, c->stream);
  HIPCHK(c, hipGetLastError());
  LegacyResult r;
  HIPCHK(c, hipMemcpyAsync(&r, c->d_legacy_r, sizeof r, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 16; i++) T[i] = r.T[i];
  *rmse = r.rmse; *found = r.found; *n_inliers = r.n_inliers;
  if (r.n_inliers > 0) {
    if (!inlier_idx || cap < r.n_inliers) return LF_ERR_CAPACITY;
    HIPCHK(c, hipMemcpy(inlier_idx, c->d_legacy_i + 2 * LF_LEGACY_CAP, sizeof(int) * (size_t)r.n_inliers, hipMemcpyDeviceToHost));
  }
  return LF_OK;
}

int lf_mle_lines(lf_ctx *c, const double *pts, const int32_t *pt_offset, const int32_t *npts, int n_lines, const double *AB_init,
                 const double K[9], lf_line_record *out, int32_t *iters) {
  if (!c || !pts || !pt_offset || !npts || !AB_init || !K || !out || n_lines < 1) return LF_ERR_INVALID;
  if (n_lines > c->fc.line_cap || n_lines > c->fc.cand_cap) return LF_ERR_CAPACITY;
  for (int i = 0; i < n_lines; i++) {
    if (npts[i] < 2 || pt_offset[i] < 0) return LF_ERR_INVALID;
    if (npts[i] > LF_MAX_SAMPLES) return LF_ERR_CAPACITY;
  }
  HIPCHK(c, hipSetDevice(c->device));
  set_camera(c, K);
  // frame slot 0: candidate i == line i == point slot i
  const int L = c->fc.line_cap;
  std::vector<double> hp((size_t)n_lines * LF_MAX_SAMPLES * 3, 0.0), ho((size_t)n_lines * LF_CAND_STRIDE, 0.0);
  std::vector<lf_line_record> hr((size_t)n_lines);
  std::vector<int> hlist((size_t)3 * L, 0);
  int cnt[3] = {0, 0, 0};
  memset(hr.data(), 0, sizeof(lf_line_record) * (size_t)n_lines);
  for (int i = 0; i < n_lines; i++) {
    memcpy(&hp[(size_t)i * LF_MAX_SAMPLES * 3], pts + 3 * (size_t)pt_offset[i], sizeof(double) * 3 * (size_t)npts[i]);
    for (int k = 0; k < 6; k++) ho[(size_t)i * LF_CAND_STRIDE + k] = AB_init[6 * (size_t)i + k];
    ho[(size_t)i * LF_CAND_STRIDE + 26] = (double)npts[i];
    hr[i].lid = i; hr[i].seg = i;
    const int which = npts[i] <= 32 ? 1 : 2;          // the work lists of k_records
    hlist[(size_t)which * L + cnt[which]++] = i;
  }
  HIPCHK(c, hipMemcpyAsync(c->fb.pts, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.cand_out, ho.data(), ho.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.recs, hr.data(), hr.size() * sizeof(lf_line_record), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.mle_list, hlist.data(), hlist.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->fb.mle_cnt, cnt, sizeof cnt, hipMemcpyHostToDevice, c->stream));
  FrontConsts fcm = c->fc;
  fcm.pts_rows = 1;                                   // candidate i == line i == row block i of fb.pts
  lf_front_launch_mle(fcm, c->fb, 1, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out, c->fb.recs, sizeof(lf_line_record) * (size_t)n_lines, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(ho.data(), c->fb.cand_out, ho.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (iters) for (int i = 0; i < n_lines; i++) iters[i] = (int)ho[(size_t)i * LF_CAND_STRIDE + 27];
  return LF_OK;
}


// ---- key-frame exchange: RCCL bound at run time --------------------------------------------------------------------------
struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static RcclApi &rccl() {
  static RcclApi a;
  static bool tried = false;
  if (tried) return a;
  tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !a.h; pass++)          // first the copy already in the process (PyTorch ships its own)
    for (const char *n : names) {
      a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (a.h) break;
    }
  if (!a.h) return a;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.h, "ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))dlsym(a.h, "ncclCommCount");
  a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.h, "ncclCommUserRank");
  a.AllGather = (decltype(a.AllGather))dlsym(a.h, "ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.h, "ncclGetErrorString");
  a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather;
  return a;
}
static int fail_nccl(lf_ctx *c, ncclResult_t r, const char *what) {
  if (c) { c->err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"); }
  return LF_ERR_HIP;
}

// pack: block (k, part) copies header / record rows of key frame k into the send buffer
__global__ void __launch_bounds__(256) k_pack_keyframes(const lf_line_record *recs, const int *nlines, const uint64_t *ids, int line_cap,
                                                         const int *slots, uint64_t id_offset, lf_line_record *send) {
  const int k = blockIdx.y, slot = slots[k];
  int n = nlines[slot];
  if (n > line_cap) n = line_cap;
  const uint4 *src = (const uint4 *)(recs + (size_t)slot * line_cap);
  uint4 *dst = (uint4 *)(send + (size_t)k * (line_cap + 1) + 1);
  const size_t words = (size_t)n * (sizeof(lf_line_record) / 16);          // 1040 = 65 x 16 bytes
  for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (size_t)gridDim.x * 256) dst[w] = src[w];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    lf_line_record *hdr = send + (size_t)k * (line_cap + 1);
    hdr->lid = nlines[slot];                     // header row: line count in `lid`, node id in the first 8 bytes of `p`
    hdr->seg = 0x4B46;                           // 'KF'
    *(uint64_t *)hdr->p = ids[slot] + id_offset;
  }
}
__global__ void k_unpack_headers(const lf_line_record *recv, int line_cap, int n_slots, int *nlines, uint64_t *ids) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const lf_line_record *hdr = recv + (size_t)s * (line_cap + 1);
  nlines[s] = hdr->lid;
  ids[s] = *(const uint64_t *)hdr->p;
}

int lf_comm_unique_id(uint8_t id[LF_COMM_ID_BYTES]) {
  if (!id) return LF_ERR_INVALID;
  if (!rccl().ok) return LF_ERR_UNSUPPORTED;
  ncclUniqueId u;
  static_assert(sizeof(u) == LF_COMM_ID_BYTES, "ncclUniqueId size");
  if (rccl().GetUniqueId(&u) != ncclSuccess) return LF_ERR_HIP;
  memcpy(id, &u, sizeof u);
  return LF_OK;
}
static void free_tracked(lf_ctx *c, void *p) {
  if (!p) return;
  for (size_t i = 0; i < c->allocs.size(); i++)
    if (c->allocs[i] == p) { c->allocs.erase(c->allocs.begin() + (long)i); break; }
  for (size_t i = 0; i < c->guards.size(); i++)      // LF_DEBUG_GUARD: the tail is checked now; lf_ctx_destroy must not read freed memory
    if (c->guards[i].p == p) { guard_check_one(c->guards[i]); c->guards.erase(c->guards.begin() + (long)i); break; }
  (void)hipFree(p);
}
static int comm_buffers(lf_ctx *c, int world, int max_kf) {
  const size_t rows = (size_t)c->fc.line_cap + 1;
  // the slot list cached on the host describes the PREVIOUS d_xslots: a communicator that is set up again starts without it
  c->h_xslots.clear();
  if (c->d_xsend && c->xbuf_world == world && c->xbuf_max_kf == max_kf) { c->comm_max_kf = max_kf; return LF_OK; }   // same geometry: keep the buffers
  if (c->d_xsend) {
    (void)hipStreamSynchronize(c->stream);
    free_tracked(c, c->d_xsend); free_tracked(c, c->d_xrecv); free_tracked(c, c->d_xnlines); free_tracked(c, c->d_xids); free_tracked(c, c->d_xslots);
    c->d_xsend = nullptr; c->d_xrecv = nullptr; c->d_xnlines = nullptr; c->d_xids = nullptr; c->d_xslots = nullptr;
  }
  c->xbuf_world = world; c->xbuf_max_kf = max_kf;
  ALLOC(c, c->d_xsend, (size_t)max_kf * rows);
  ALLOC(c, c->d_xrecv, (size_t)world * max_kf * rows);
  ALLOC(c, c->d_xnlines, (size_t)world * max_kf);
  ALLOC(c, c->d_xids, (size_t)world * max_kf);
  ALLOC(c, c->d_xslots, (size_t)max_kf);
  HIPCHK(c, hipMemsetAsync(c->d_xsend, 0, (size_t)max_kf * rows * sizeof(lf_line_record), c->stream));   // header rows: unused bytes are zero
  c->comm_max_kf = max_kf;
  return LF_OK;
}
int lf_comm_init(lf_ctx *c, int world_size, int rank, const uint8_t id[LF_COMM_ID_BYTES], int max_keyframes) {
  if (!c || !id || world_size < 1 || rank < 0 || rank >= world_size || max_keyframes < 1) return LF_ERR_INVALID;
  if (c->comm) return LF_ERR_INVALID;
  if (!rccl().ok) { c->err = "librccl.so.1 not found"; return LF_ERR_UNSUPPORTED; }
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclResult_t r = rccl().CommInitRank(&c->comm, world_size, u, rank);
  if (r != ncclSuccess) { c->comm = nullptr; return fail_nccl(c, r, "ncclCommInitRank"); }
  c->comm_owner = true; c->comm_world = world_size; c->comm_rank = rank;
  return comm_buffers(c, world_size, max_keyframes);
}
int lf_comm_attach(lf_ctx *c, lf_ctx *owner) {
  if (!c || !owner || !owner->comm || c->comm || c->device != owner->device || c->fc.line_cap != owner->fc.line_cap) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  c->comm = owner->comm; c->comm_owner = false; c->comm_world = owner->comm_world; c->comm_rank = owner->comm_rank;
  return comm_buffers(c, owner->comm_world, owner->comm_max_kf);
}
int lf_comm_destroy(lf_ctx *c) {
  if (!c) return LF_ERR_INVALID;
  if (c->comm && c->comm_owner) {
    (void)hipStreamSynchronize(c->stream);
    (void)rccl().CommDestroy(c->comm);
  }
  c->comm = nullptr; c->comm_owner = false;
  c->h_xslots.clear();
  return LF_OK;
}
int lf_ctx_device_bytes(lf_ctx *c, unsigned long long *held, unsigned long long *dev_free, unsigned long long *dev_total) {
  if (!c) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  size_t fr = 0, tot = 0;
  HIPCHK(c, hipMemGetInfo(&fr, &tot));
  if (held) *held = c->alloc_bytes;
  if (dev_free) *dev_free = fr;
  if (dev_total) *dev_total = tot;
  return LF_OK;
}
int lf_comm_info(lf_ctx *c, int *n_ranks, int *rank, long long *n_allgathers) {
  if (!c) return LF_ERR_INVALID;
  if (!c->comm) { c->err = "lf_comm_init has not been called"; return LF_ERR_INVALID; }
  if (!rccl().CommCount || !rccl().CommUserRank) return LF_ERR_UNSUPPORTED;
  int n = 0, r = 0;
  ncclResult_t e = rccl().CommCount(c->comm, &n);
  if (e != ncclSuccess) return fail_nccl(c, e, "ncclCommCount");
  e = rccl().CommUserRank(c->comm, &r);
  if (e != ncclSuccess) return fail_nccl(c, e, "ncclCommUserRank");
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  if (n_allgathers) *n_allgathers = c->n_allgathers;
  return LF_OK;
}
int lf_allgather_keyframes(lf_ctx *c, const int32_t *kf_slots, int n_kf, uint64_t id_offset, const lf_line_record **d_recs,
                           const int32_t **d_nlines, const uint64_t **d_ids, int *n_frames, int *ext_line_cap) {
  if (!c || !kf_slots || n_kf < 1 || !d_recs || !d_nlines || !d_ids || !n_frames || !ext_line_cap) return LF_ERR_INVALID;
  if (!c->comm) { c->err = "lf_comm_init has not been called"; return LF_ERR_INVALID; }
  if (n_kf > c->comm_max_kf) return LF_ERR_CAPACITY;
  for (int k = 0; k < n_kf; k++) if (kf_slots[k] < 0 || kf_slots[k] >= c->last_batch) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->h_xslots.size() != (size_t)n_kf || memcmp(c->h_xslots.data(), kf_slots, sizeof(int) * (size_t)n_kf) != 0) {
    // a new slot list: upload it and wait (the caller's array may be a temporary).  The same list again -- the usual case,
    // every step of a run -- costs nothing and, above all, no host synchronisation: the exchange stays asynchronous.
    c->h_xslots.assign(kf_slots, kf_slots + n_kf);
    HIPCHK(c, hipMemcpyAsync(c->d_xslots, c->h_xslots.data(), sizeof(int) * (size_t)n_kf, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  const int L = c->fc.line_cap;
  hipLaunchKernelGGL(k_pack_keyframes, dim3(16, n_kf), dim3(256), 0, c->stream, (const lf_line_record *)c->fb.recs, (const int *)c->fb.nlines,
                     (const uint64_t *)c->d_frame_ids, L, (const int *)c->d_xslots, id_offset, c->d_xsend);
  HIPCHK(c, hipGetLastError());
  const size_t bytes = (size_t)n_kf * (L + 1) * sizeof(lf_line_record);
  ncclResult_t r = rccl().AllGather(c->d_xsend, c->d_xrecv, bytes, ncclUint8, c->comm, c->stream);   // the ONE collective
  if (r == ncclSuccess) c->n_allgathers++;
  if (r != ncclSuccess) return fail_nccl(c, r, "ncclAllGather");
  const int ns = c->comm_world * n_kf;
  hipLaunchKernelGGL(k_unpack_headers, dim3((ns + 255) / 256), dim3(256), 0, c->stream, (const lf_line_record *)c->d_xrecv, L, ns, c->d_xnlines, c->d_xids);
  HIPCHK(c, hipGetLastError());
  *d_recs = c->d_xrecv + 1;                 // the records of slot s start one (header) row into its block
  *d_nlines = c->d_xnlines; *d_ids = c->d_xids; *n_frames = ns; *ext_line_cap = L + 1;
  return LF_OK;
}


// ---- ORB extractor ---------------------------------------------------------------------------------------------------
static int orb_prepare(lf_ctx *c) {
  OrbConsts &o = c->oc;
  memset(&o, 0, sizeof o);
  o.W = c->W; o.H = c->H;
  int off = 0;
  for (int l = 0; l < LF_ORB_LEVELS; l++) {
    const float sf = (float)pow((double)1.2f, (double)l);          // getScale (aorb.cpp:555-558), float scaleFactor member
    const float inv = 1 / sf;
    o.sf[l] = sf; o.inv_sf[l] = inv;
    o.lw[l] = (int)nearbyint((double)((float)c->W * inv));          // Size(cvRound(cols * scale), cvRound(rows * scale))
    o.lh[l] = (int)nearbyint((double)((float)c->H * inv));
    if (o.lw[l] < 1 || o.lh[l] < 1 || o.lw[l] > 1023 || o.lh[l] > 511) return LF_ERR_UNSUPPORTED;   // key layout: x 10 bits, y 9 bits
    o.loff[l] = off;
    off += o.lw[l] * o.lh[l];
    if (l > 0) { o.scale_x[l] = 1. / ((double)o.lw[l] / o.lw[l - 1]); o.scale_y[l] = 1. / ((double)o.lh[l] / o.lh[l - 1]); }
  }
  o.total = off;
  {   // nfeaturesPerLevel (aorb.cpp:612-624) for nfeatures = 10000
    const int nfeatures = 10000;
    float factor = (float)(1.0 / (double)1.2f);
    float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)LF_ORB_LEVELS));
    int sum = 0;
    for (int l = 0; l < LF_ORB_LEVELS - 1; l++) { o.nper[l] = (int)nearbyint((double)nd); sum += o.nper[l]; nd *= factor; }
    o.nper[LF_ORB_LEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  }
  {   // umax (aorb.cpp:632-647)
    int v, v0, vmax = (int)floor(LF_ORB_HALF * sqrt(2.f) / 2 + 1), vmin = (int)ceil(LF_ORB_HALF * sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) o.umax[v] = (int)nearbyint(sqrt((double)LF_ORB_HALF * LF_ORB_HALF - v * v));
    for (v = LF_ORB_HALF, v0 = 0; v >= vmin; --v) {
      while (o.umax[v0] == o.umax[v0 + 1]) ++v0;
      o.umax[v] = v0;
      ++v0;
    }
  }
  {   // cv::getGaussianKernel(7, 2, CV_32F) -> 8-bit fixed point (filter.cpp): host libm exp, as the reference's CPU does
    double s2 = -0.5 / (2.0 * 2.0), sum = 0;
    float cf[7];
    for (int i = 0; i < 7; i++) { double x = i - 3.0, t = exp(s2 * x * x); cf[i] = (float)t; sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); o.blur_k[i] = (int)nearbyint((double)(cf[i] * 256.f)); }
  }
  const size_t B = (size_t)c->maxB;
  OrbBuffers &b = c->ob;
  memset(&b, 0, sizeof b);
  ALLOC(c, b.pyr, B * o.total); ALLOC(c, b.blur, B * o.total); ALLOC(c, b.score, B * o.total);
  ALLOC(c, b.cand, B * LF_ORB_CAND_CAP); ALLOC(c, b.ncand, B); ALLOC(c, b.hist, B * LF_ORB_LEVELS * 256);
  ALLOC(c, b.sel, B * LF_ORB_KP_MAX * 4); ALLOC(c, b.nsel, B);
  ALLOC(c, c->d_orb_nkp, 2 * B);
  ALLOC(c, c->d_orb_thr, B);
  ALLOC(c, c->d_orb_adj, 1);
  c->orb_ready = true;
  return LF_OK;
}

int lf_orb_extract_device(lf_ctx *c, const uint8_t *d_gray, size_t gray_frame_stride, int gray_row_stride, const float *d_depth,
                          size_t depth_frame_stride, int depth_row_stride, int n_frames, int fast_threshold, int max_keypoints,
                          float *d_kp_xy, float *d_kp_meta, uint8_t *d_desc, int32_t *d_nkp, int kp_cap) {
  if (!c || !d_gray || !d_kp_xy || !d_desc || !d_nkp || n_frames < 1 || gray_row_stride < c->W || kp_cap < 1 || max_keypoints < 1 ||
      (d_depth && depth_row_stride < c->W))
    return LF_ERR_INVALID;
  if (n_frames > c->maxB) return LF_ERR_CAPACITY;
  if (max_keypoints > LF_ORB_KP_MAX) return LF_ERR_UNSUPPORTED;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->orb_ready) { int r = orb_prepare(c); if (r != LF_OK) return r; }
  OrbConsts oc = c->oc;
  oc.fast_threshold = fast_threshold; oc.max_keypoints = max_keypoints; oc.kp_cap = kp_cap;
  OrbBuffers ob = c->ob;
  ob.gray = d_gray; ob.gray_frame_stride = gray_frame_stride; ob.gray_row_stride = gray_row_stride;
  ob.depth = d_depth; ob.depth_frame_stride = depth_frame_stride; ob.depth_row_stride = depth_row_stride;
  ob.kp_xy = d_kp_xy; ob.kp_meta = d_kp_meta; ob.desc = d_desc; ob.nkp = c->d_orb_nkp;
  hipStream_t pst = pt_stream_begin(c);
  lf_orb_launch(oc, ob, n_frames, pst);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(d_nkp, c->ob.nsel, sizeof(int) * (size_t)n_frames, hipMemcpyDeviceToDevice, pst));
  c->orb_last = n_frames;
  return LF_OK;
}

void lf_orb_adjuster_init(lf_orb_adjuster *a, int max_keypoints, int max_iters) {
  if (!a) return;
  a->thresh = 20.0; a->min_thresh = 2.0; a->max_thresh = 10000.0;          // DetectorAdjuster("AORB", 20), features.cpp:75-76, feature_adjuster.h:15
  a->increase_factor = 1.3; a->decrease_factor = 0.7;
  a->min_features = max_keypoints;                                            // features.cpp:88-90: "shall not get below max"
  a->max_features = (int)(max_keypoints * 1.5);
  a->max_iters = max_iters;
}
int lf_orb_extract_adjusted_device(lf_ctx *c, const uint8_t *d_gray, size_t gray_frame_stride, int gray_row_stride, const float *d_depth,
                                   size_t depth_frame_stride, int depth_row_stride, int n_frames, const lf_orb_adjuster *adj, int reset_state,
                                   int max_keypoints, float *d_kp_xy, float *d_kp_meta, uint8_t *d_desc, int32_t *d_nkp, int kp_cap,
                                   int32_t *d_thresholds_out) {
  if (!c || !d_gray || !d_kp_xy || !d_desc || !d_nkp || !adj || n_frames < 1 || gray_row_stride < c->W || kp_cap < 1 || max_keypoints < 1 ||
      (d_depth && depth_row_stride < c->W))
    return LF_ERR_INVALID;
  if (!(adj->min_thresh >= 1.0) || !(adj->max_thresh > adj->min_thresh) || !(adj->increase_factor > 1.0) || !(adj->decrease_factor > 0.0) ||
      !(adj->decrease_factor < 1.0) || adj->max_iters < 1 || adj->min_features < 0 || adj->max_features < adj->min_features)
    return LF_ERR_INVALID;
  if (n_frames > c->maxB) return LF_ERR_CAPACITY;
  if (max_keypoints > LF_ORB_KP_MAX) return LF_ERR_UNSUPPORTED;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->orb_ready) { int r = orb_prepare(c); if (r != LF_OK) return r; }
  hipStream_t pst = pt_stream_begin(c);
  if (reset_state || !c->orb_adj_set) {
    c->h_orb_adj = adj->thresh;                             // (a member: the copy is asynchronous)
    HIPCHK(c, hipMemcpyAsync(c->d_orb_adj, &c->h_orb_adj, sizeof(double), hipMemcpyHostToDevice, pst));
    c->orb_adj_set = true;
  }
  OrbConsts oc = c->oc;
  oc.max_keypoints = max_keypoints; oc.kp_cap = kp_cap;
  OrbBuffers ob = c->ob;
  ob.gray = d_gray; ob.gray_frame_stride = gray_frame_stride; ob.gray_row_stride = gray_row_stride;
  ob.depth = d_depth; ob.depth_frame_stride = depth_frame_stride; ob.depth_row_stride = depth_row_stride;
  ob.kp_xy = d_kp_xy; ob.kp_meta = d_kp_meta; ob.desc = d_desc; ob.nkp = c->d_orb_nkp;
  ob.thr_frame = c->d_orb_thr; ob.adj_state = c->d_orb_adj;
  OrbAdjuster a;
  a.min_thresh = adj->min_thresh; a.max_thresh = adj->max_thresh; a.inc = adj->increase_factor; a.dec = adj->decrease_factor;
  a.min_features = adj->min_features; a.max_features = adj->max_features; a.max_iters = adj->max_iters;
  a.base_threshold = (int)adj->min_thresh;                  // the lowest threshold a detection can be made with
  lf_orb_launch_adjusted(oc, ob, a, n_frames, pst);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(d_nkp, c->ob.nsel, sizeof(int) * (size_t)n_frames, hipMemcpyDeviceToDevice, pst));
  if (d_thresholds_out) HIPCHK(c, hipMemcpyAsync(d_thresholds_out, c->d_orb_thr, sizeof(int) * (size_t)n_frames, hipMemcpyDeviceToDevice, pst));
  c->orb_last = n_frames;
  return LF_OK;
}
int lf_orb_adjuster_state(lf_ctx *c, double *thresh) {
  if (!c || !thresh || !c->orb_ready || !c->orb_adj_set) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  pt_stream_join(c);
  HIPCHK(c, hipMemcpyAsync(thresh, c->d_orb_adj, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LF_OK;
}

int lf_orb_check(lf_ctx *c) {
  if (!c || !c->orb_ready || c->orb_last < 1) return LF_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  pt_stream_join(c);
  std::vector<int> h((size_t)2 * c->orb_last), ns((size_t)c->orb_last);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->d_orb_nkp, sizeof(int) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(ns.data(), c->ob.nsel, sizeof(int) * ns.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int f = 0; f < c->orb_last; f++)
    if (h[(size_t)c->orb_last + f] != 0 || h[f] > ns[f]) { c->err = "ORB: more corners / key points in a frame than the buffers hold"; return LF_ERR_CAPACITY; }
  return LF_OK;
}

int lf_orb_get_level(lf_ctx *c, int frame, int level, int blurred, uint8_t *out, size_t cap_bytes, int *w, int *h) {
  if (!c || !c->orb_ready || frame < 0 || frame >= c->orb_last || level < 0 || level >= LF_ORB_LEVELS) return LF_ERR_INVALID;
  if (w) *w = c->oc.lw[level];
  if (h) *h = c->oc.lh[level];
  const size_t bytes = (size_t)c->oc.lw[level] * c->oc.lh[level];
  if (!out) return LF_OK;
  if (cap_bytes < bytes) return LF_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  pt_stream_join(c);
  const uint8_t *src = (blurred ? c->ob.blur : c->ob.pyr) + (size_t)frame * c->oc.total + c->oc.loff[level];
  HIPCHK(c, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LF_OK;
}


// ---- EDLines ---------------------------------------------------------------------------------------------------------
// nfa() of the reference's libEDLines.a (NFA.o; LSD's: log-gamma by Lanczos / Windschitl, binomial tail with 10 % tolerance,
// reciprocals of the term index) -- evaluated on the host with libm, once per image size, exactly as the binary builds its
// NFALUT; the device sees the table only.
static double ed_log_gamma(double x) {
  if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * log(x + 5.5) - (x + 5.5), b = 0.0;
  for (int n = 0; n < 7; n++) { a -= log(x + (double)n); b += q[n] * pow(x, (double)n); }
  return a + log(b);
}
static double ed_nfa(int n, int k, double p, double logNT) {
  const double tolerance = 0.1;
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * log10(p);
  const double p_term = p / (1.0 - p);
  const double log1term = ed_log_gamma((double)n + 1.0) - ed_log_gamma((double)k + 1.0) - ed_log_gamma((double)(n - k) + 1.0) + (double)k * log(p) +
                          (double)(n - k) * log(1.0 - p);
  double term = exp(log1term);
  if (term == 0.0) {
    if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
    return -logNT;
  }
  double bin_tail = term;
  for (int i = k + 1; i <= n; i++) {
    const double bin_term = (double)(n - i + 1) * (1.0 / (double)i);
    const double mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1.0) {
      const double err = term * ((1.0 - pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < tolerance * fabs(-log10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  return -log10(bin_tail) - logNT;
}
static int ed_prepare(lf_ctx *c) {
  EdConsts &e = c->ec;
  memset(&e, 0, sizeof e);
  e.W = c->W; e.H = c->H;
  if (c->W > 65535 || c->H > 65535 || c->W < 8 || c->H < 8) return LF_ERR_UNSUPPORTED;   // pixels are packed as row << 16 | column
  const double logNT = 2.0 * (log10((double)c->W) + log10((double)c->H));
  {   // ComputeMinLineLength: Round(logNT / -log10(1/8) * 0.5), at least 9
    const int n = (int)floor(logNT / 0.90308998699194354 * 0.5 + 0.5);
    e.min_len = n < 9 ? 9 : n;
  }
  e.lut_size = (c->W + c->H) / 8;
  e.nmax = 4 * (c->W + c->H) + 8;            // EnumerateRectPoints stops after 4 (|dx| + |dy|) pixels
  e.seg_cap = c->lc.seg_cap;
  e.anchor_cap = c->W * c->H / 2;
  e.segtab_cap = c->W * c->H / 10 + 16;
  std::vector<int> kmin((size_t)e.nmax);
  {   // NFALUT(size, 0.125, logNT) of the binary for n < size; beyond it the binary evaluates nfa(n, k) >= 0 directly, which is
      // k >= the smallest such k (the tail decreases with k)
    const int size = e.lut_size;
    int j = 1;
    kmin[0] = 1;
    for (int i = 1; i < size && i < e.nmax; i++) {
      kmin[(size_t)i] = size + 1;
      double ret = ed_nfa(i, j, 0.125, logNT);
      if (ret < 0) {
        while (j < i) { j++; ret = ed_nfa(i, j, 0.125, logNT); if (ret >= 0) break; }
        if (ret < 0) continue;
      }
      kmin[(size_t)i] = j;
    }
    for (int i = (size > 1 ? size : 1); i < e.nmax; i++) {
      int k = 0;
      while (k <= i && !(ed_nfa(i, k, 0.125, logNT) >= 0.0)) k++;
      kmin[(size_t)i] = k;      // (i + 1: never)
    }
  }
  std::vector<double> lut(1025);
  for (int i = 0; i <= 1024; i++) lut[(size_t)i] = atan((double)i * (1.0 / 1024.0));   // myAtan2's table
  const size_t B = (size_t)c->maxB, HW = (size_t)c->W * c->H, WH8 = (size_t)(c->W + c->H) * 8;
  EdBuffers &b = c->eb;
  memset(&b, 0, sizeof b);
  int *d_kmin = nullptr;
  double *d_lut = nullptr;
  ALLOC(c, b.smooth, B * HW); ALLOC(c, b.D, B * HW); ALLOC(c, b.E, B * HW); ALLOC(c, b.G, B * HW);
  ALLOC(c, b.hist, B * LF_ED_BINS); ALLOC(c, b.anchors, B * (size_t)e.anchor_cap); ALLOC(c, b.nanch, B);
  ALLOC(c, b.walk, B * HW); ALLOC(c, b.stack, B * (size_t)LF_ED_STACK_CAP * 2); ALLOC(c, b.chains, B * (size_t)(LF_ED_CHAIN_CAP + 1));
  ALLOC(c, b.chain_nos, B * WH8); ALLOC(c, b.segpix, B * HW); ALLOC(c, b.segtab, B * (size_t)e.segtab_cap * 2);
  ALLOC(c, b.lines, B * (size_t)LF_ED_LINE_CAP);
  ALLOC(c, b.nsegtab, B); ALLOC(c, b.seg_nl, B * (size_t)e.segtab_cap * 3); ALLOC(c, b.nslots, B); ALLOC(c, b.lvalid, B * (size_t)LF_ED_LINE_CAP);
  ALLOC(c, d_kmin, kmin.size()); ALLOC(c, d_lut, lut.size());
  HIPCHK(c, hipMemcpyAsync(d_kmin, kmin.data(), kmin.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_lut, lut.data(), lut.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  b.kmin = d_kmin; b.atan_lut = d_lut;
  b.segs = c->lb.segs; b.nsegs = c->lb.nsegs;
  c->ed_ready = true;
  return LF_OK;
}

int lf_edlines_batch_device(lf_ctx *c, const uint8_t *d_gray, size_t frame_stride, int row_stride, int n_frames) {
  if (!c || !d_gray || n_frames < 1 || row_stride < c->W) return LF_ERR_INVALID;
  if (n_frames > c->maxB) return LF_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->ed_ready) { int r = ed_prepare(c); if (r != LF_OK) return r; }
  EdBuffers b = c->eb;
  b.gray = d_gray; b.gray_frame_stride = frame_stride; b.gray_row_stride = row_stride;
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  lf_edlines_launch(c->ec, b, n_frames, c->stream);
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));        // (stage timers: the whole detector counts as stage 0, stage 1 is empty)
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  HIPCHK(c, hipGetLastError());
  c->last_batch = n_frames;
  return LF_OK;
}

}  // extern "C"
