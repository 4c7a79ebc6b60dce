// lf_points.hip -- the point side that feeds the hybrid solver (without the ORB extractor):
//   k_project3d   Node::projectTo3D       (src/node.cpp:952-1018)  key points + depth -> feature_locations_3d_
//   k_featmatch   Node::featureMatching   (src/node.cpp:606-641, BRUTEFORCE / ORB branch): Hamming k = 2 nearest
//                 neighbours (lower train index wins ties, as OpenCV's batchDistance), ratio test, unique-train filter
//                 in query order, distance offset from the counter-based generator
// Sequential twins: oracle/point_oracle.c.
#include "lf_points.h"
#include "lf_linalg.h"

typedef unsigned long long u64;
#define LF_STREAM_FEAT(fq, ft) ((((uint64_t)(fq)) << 32) ^ (uint64_t)(uint32_t)(ft) ^ 0x4000000000000000ULL)

// one wavefront per frame: ordered compaction of the valid key points, stop at max_keyp (node.cpp:985-1013)
__global__ void __launch_bounds__(64) k_project3d(PointConsts c, PointBuffers b) {
  const int f = blockIdx.x, lane = (int)(threadIdx.x & 63u);
  const float *depth = b.depth + (size_t)f * b.depth_frame_stride;
  const float *kp = b.kp_xy + (size_t)f * c.kp_cap * 2;
  float *out = b.points + (size_t)f * c.kp_cap * 4;
  int *kept = b.kept ? b.kept + (size_t)f * c.kp_cap : nullptr;
  int n = b.nkp[f];
  if (n > c.kp_cap) n = c.kp_cap;
  const float fx = (float)(1. / c.K[0]), fy = (float)(1. / c.K[4]), cx = (float)c.K[2], cy = (float)c.K[5];
  int m = 0;
  for (int base = 0; base < n && m < c.max_keyp; base += 64) {
    int i = base + lane;
    bool ok = false;
    float px = 0, py = 0, Z = 0;
    if (i < n) {
      px = kp[2 * i]; py = kp[2 * i + 1];
      if (!(px >= c.W || px < 0 || py >= c.H || py < 0 || px != px || py != py)) {
        int iy = (int)__builtin_round((double)py), ix = (int)__builtin_round((double)px);
        if (iy > c.H - 1) iy = c.H - 1;
        if (ix > c.W - 1) ix = c.W - 1;
        Z = (float)(depth[(size_t)iy * b.depth_row_stride + ix] * c.depth_scaling);
        ok = (Z == Z);
      }
    }
    u64 msk = __ballot(ok);
    int at = m + __popcll(msk & ((1ull << lane) - 1ull));
    if (ok && at < c.max_keyp) {
      float x = (px - cx) * Z * fx, y = (py - cy) * Z * fy;
      out[4 * at] = x; out[4 * at + 1] = y; out[4 * at + 2] = Z; out[4 * at + 3] = 1.0f;
      if (kept) kept[at] = i;
    }
    m += __popcll(msk);
  }
  if (m > c.max_keyp) m = c.max_keyp;
  if (lane == 0) b.npts[f] = m;
}

#define FM_THREADS 256
// one 256-thread block per node pair; train descriptors staged in LDS
__global__ void __launch_bounds__(FM_THREADS) k_featmatch(PointConsts c, PointBuffers b) {
  extern __shared__ uint32_t fm_lds[];   // [cap][8] train descriptors | [cap] owner | [4] wave counts
  const int pr = blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fq = b.pair_q[pr], ft = b.pair_t[pr];
  uint32_t *tdesc = fm_lds;
  int *owner = (int *)(fm_lds + (size_t)c.desc_cap * 8);
  int *wcnt = owner + c.desc_cap;
  int nq = b.ndesc[fq], nt = b.ndesc[ft];
  if (nq > c.desc_cap) nq = c.desc_cap;
  if (nt > c.desc_cap) nt = c.desc_cap;
  const uint32_t *gq = (const uint32_t *)(b.desc + (size_t)fq * c.desc_cap * 32);
  const uint32_t *gt = (const uint32_t *)(b.desc + (size_t)ft * c.desc_cap * 32);
  for (int i = tid; i < nt * 8; i += FM_THREADS) tdesc[i] = gt[i];
  for (int i = tid; i < nt; i += FM_THREADS) owner[i] = 1 << 30;
  __syncthreads();
  int *oq = b.fm_q + (size_t)pr * c.desc_cap, *ot = b.fm_t + (size_t)pr * c.desc_cap;
  float *od = b.fm_d + (size_t)pr * c.desc_cap;
  const uint64_t stream = LF_STREAM_FEAT(b.frame_ids[fq], b.frame_ids[ft]);
  const int rounds = (nq + FM_THREADS - 1) / FM_THREADS;
  // pass 1: two nearest neighbours + ratio test; the first query (lowest index) that claims a train point owns it
  // (the claims of later rounds cannot beat those of earlier rounds: atomicMin on the query index)
  int b1r[4];     // desc_cap <= 1024 -> at most 4 rounds
  float ratio[4];
  for (int r = 0; r < rounds && r < 4; r++) {
    int i = r * FM_THREADS + tid;
    b1r[r] = -1; ratio[r] = 0;
    if (i < nq && nt >= 2) {
      uint32_t q[8];
#pragma unroll
      for (int k = 0; k < 8; k++) q[k] = gq[(size_t)i * 8 + k];
      int b1 = -1, d1 = 1 << 30, d2 = 1 << 30;
      for (int j = 0; j < nt; j++) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __popc(q[k] ^ tdesc[j * 8 + k]);
        if (d < d1) { d2 = d1; d1 = d; b1 = j; }
        else if (d < d2) d2 = d;
      }
      float fr = (float)d1 / (float)d2;
      if ((double)fr < c.nn_ratio) { b1r[r] = b1; ratio[r] = fr; atomicMin(&owner[b1], i); }
    }
  }
  __syncthreads();
  // pass 2: ordered emission (query order)
  int total = 0;
  for (int r = 0; r < rounds && r < 4; r++) {
    int i = r * FM_THREADS + tid;
    bool keep = b1r[r] >= 0 && owner[b1r[r]] == i;
    u64 m = __ballot(keep);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; w++) off += wcnt[w];
    if (keep) {
      int at = off + __popcll(m & ((1ull << lane) - 1ull));
      oq[at] = i; ot[at] = b1r[r];
      od[at] = (float)(ratio[r] + (float)lf_rand31(c.rng_seed, stream, (uint64_t)i) / (1000.0 * 2147483647.0));
    }
    total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  if (tid == 0) b.fm_n[pr] = total;
}

// OpenNIListener::loadRawData pixel conversions (src/openni_listener.cpp:1233-1246) + the grey image of Node::Node
// (src/node.cpp:193: cvtColor(CV_RGB2GRAY) applied to the BGR matrix of cv::imread, i.e. memory channel 0 = BLUE is
// weighted as red): gray = (4899 B + 9617 G + 1868 R + 8192) >> 14 (OpenCV 2.4 fixed point), depth = u16 -> float,
// values < 1e-5 -> NaN, times (float)(1/5000.0).
__global__ void __launch_bounds__(256) k_ingest_tum(const uint8_t *rgb, const uint16_t *depth16, uint8_t *gray, float *depth,
                                                    size_t npix, float inv_factor) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  unsigned r = rgb[3 * i], g = rgb[3 * i + 1], bl = rgb[3 * i + 2];
  gray[i] = (uint8_t)((bl * 4899u + g * 9617u + r * 1868u + 8192u) >> 14);
  float d = (float)depth16[i];
  if (d < 1e-5) d = __builtin_nanf("");
  depth[i] = d * inv_factor;
}
void lf_points_ingest_launch(const uint8_t *rgb, const uint16_t *depth16, uint8_t *gray, float *depth, size_t npix,
                             double depth_factor, hipStream_t st) {
  hipLaunchKernelGGL(k_ingest_tum, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, rgb, depth16, gray, depth, npix,
                     (float)(1.0 / depth_factor));
}

void lf_points_project_launch(const PointConsts &c, const PointBuffers &b, int n_frames, hipStream_t st) {
  hipLaunchKernelGGL(k_project3d, dim3(n_frames), dim3(64), 0, st, c, b);
}
void lf_points_match_launch(const PointConsts &c, const PointBuffers &b, int n_pairs, hipStream_t st) {
  size_t lds = (size_t)c.desc_cap * 8 * 4 + (size_t)c.desc_cap * 4 + 16;
  hipLaunchKernelGGL(k_featmatch, dim3(n_pairs), dim3(FM_THREADS), lds, st, c, b);
}
