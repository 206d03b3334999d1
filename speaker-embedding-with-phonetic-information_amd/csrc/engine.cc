// Device engine.  See engine.h.
#include "engine.h"
#include "knobs.h"

#include <stddef.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <exception>
#include <sstream>
#include <thread>

namespace xv {

namespace {

constexpr int kDefaultFastMinPooledMx2 = 160;   // the deeper c-vector network measured 7.9e-5 at 117 pooled frames (heavy-tailed model)
constexpr int kHalo = 32;  // zero rows in front of / behind every frame-level plane (|offset| <= 15)
constexpr uint32_t kBlobVersion = 8;   // 3: one E8M0 scale per (row, block of four K steps, lane group) of the residual
                                       // plane; 4: + the 4-bit weight image of kPrecFp16Mx2; 5: + the residual plane in the
                                       // K-walk order of tdnn_gemm_kernel_p8; 6: + the 4-bit weight image in the order of ITS second walk;
                                       // 7: that image only for layers without time offsets (the ones the kernel runs in 1.5 passes)
                                       // 8: reserved[0..1] of the header = fingerprint of the image (what a calibration file names)
constexpr uint64_t kNone = ~0ull;

// FNV-1a over 64-bit words (the tail byte-wise): a content fingerprint, not a cryptographic hash
uint64_t Hash64(const uint8_t* p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
  }
  for (; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
  return h ? h : 1;   // 0 = "unknown"
}

struct BlobHeader {
  char magic[8];
  uint32_t version;
  int32_t precision;
  int32_t input_dim, n_layers, pooled_layer, pool_dim, pool_left, pool_right;
  float variance_floor;
  int32_t output_layer, output_dim, output_is_segment, left_context, right_context, min_frames;
  int32_t reserved[7];
  uint64_t data_offset;
  uint64_t total_bytes;
};

struct BlobLayer {
  char name[64];
  int32_t nsrc;
  int32_t src_layer[kMaxSeg], src_offset[kMaxSeg], src_dim[kMaxSeg];
  int32_t in_dim, out_dim, k_pad, n_pad, relu, bn, log_softmax, segment_level, left, right;
  uint64_t w_hi, w_lo, bias, scale, offset;  // relative to data_offset
  uint64_t w4, w4_scale;                     // kPrecFp16Mx residual plane + its E8M0 scales in both tile orders, or kNone
  int32_t ldw4, ldw4b;
  uint64_t w4b, w4b_scale;                   // kPrecFp16Mx2: 4-bit image of the weights + scales (both tile orders), or kNone
  uint64_t w4p, w4p_scale;                   // the residual plane of w4 in the walk order of tdnn_gemm_kernel_p8 (PlanWalkSteps64:
                                             // 64-column chunk -> offset) + scales, for the layers that kernel can run, or kNone
  uint64_t w4bp, w4bp_scale;                 // kPrecFp16Mx2 on that kernel: the 4-bit weight image in the order of its second walk
                                             // (PlanWalkLoSteps64: tiles of 256 columns), rows of k_pad * 2 bytes (the pitch of
                                             // the fp16 plane; a row's tiles fill the first quarter), + scales, or kNone
};

// Round-to-nearest-even onto the e2m1 grid {0, .5, 1, 1.5, 2, 3, 4, 6} (saturating), like v_cvt_scalef32_pk_fp4_*.
inline uint8_t ToE2M1(float x) {
  static const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  const float a = std::fabs(x);
  int best = 7;
  if (!(a >= 6.f)) {
    best = 0;
    for (int i = 1; i < 8; ++i) {
      const float dl = a - grid[best], dh = grid[i] - a;
      if (dh < dl || (dh == dl && (i & 1) == 0)) best = i;
      if (grid[i] >= a) break;
    }
  }
  return (uint8_t)(best | (std::signbit(x) ? 8 : 0));
}

inline int RoundUp(int x, int m) { return (x + m - 1) / m * m; }
// K columns a source occupies in a layer's weight planes.  The modes with 4-bit products walk K in blocks of 128 columns:
// there a source that is another layer's output is padded to a multiple of 128 (its plane is that wide and zero beyond
// the layer's dimension, the weights of the padding are zero) - a 650-wide source (the phonetic branch of the c-vector
// networks) costs 14 % more steps and can then run them instead of the two-pass arithmetic.
inline int SrcKPad(int dim, int src_layer, bool segment_level, int precision) {
  const bool blocks = precision == kPrecAuto || precision == kPrecFp16Mx || precision == kPrecFp16Mx2;
  return RoundUp(dim, (blocks && !segment_level && src_layer >= 0) ? 128 : kBK);
}
inline uint64_t Align256(uint64_t x) { return (x + 255) & ~255ull; }

}  // namespace

void PackMxRow(const float* res, int k_pad, const int* step_wcol, uint8_t* row, uint8_t* scales) {
  const int nblk = k_pad / kBK / 4;
  memset(row, 0, (size_t)nblk * 64);
  for (int blk = 0; blk < nblk; ++blk)
    for (int g = 0; g < 4; ++g) {
      // the 32 values lane group g holds of this block: columns 8 g .. 8 g + 7 of each of its four steps
      float mx = 0.f;
      for (int e = 0; e < 32; ++e) mx = std::max(mx, std::fabs(res[step_wcol[4 * blk + e / 8] + 8 * g + e % 8]));
      int e8 = 127;
      if (mx > 0.f && std::isfinite(mx)) {
        // smallest power of two 2^E with max |r| / 2^E <= 6 (the largest e2m1 value)
        int ex;
        const float m = frexpf(mx / 6.f, &ex);   // mx / 6 = m * 2^ex, m in [0.5, 1)
        const int E = (m == 0.5f) ? ex - 1 : ex;
        e8 = std::min(std::max(127 + E, 1), 254);
      }
      scales[4 * blk + g] = (uint8_t)e8;
      const float inv_s = ldexpf(1.f, 127 - e8);
      for (int e = 0; e < 32; ++e) {
        const int col = step_wcol[4 * blk + e / 8] + 8 * g + e % 8;
        row[blk * 64 + g * 16 + e / 2] |= (uint8_t)(ToE2M1(res[col] * inv_s) << (4 * (e & 1)));
      }
    }
}

void PackMxWeightsRow(const float* w, const int* lo_wcol, int n_lo, uint8_t* row, uint8_t* scales) {
  memset(row, 0, (size_t)n_lo * 64);
  for (int t = 0; t < n_lo; ++t)
    for (int g = 0; g < 4; ++g) {
      const float* c = w + lo_wcol[t] + 32 * g;   // the 32 consecutive columns lane group g holds of this step
      float mx = 0.f;
      for (int e = 0; e < 32; ++e) mx = std::max(mx, std::fabs(c[e]));
      int e8 = 127;
      if (mx > 0.f && std::isfinite(mx)) {
        int ex;
        const float m = frexpf(mx / 6.f, &ex);
        const int E = (m == 0.5f) ? ex - 1 : ex;
        e8 = std::min(std::max(127 + E, 1), 254);
      }
      scales[4 * t + g] = (uint8_t)e8;
      const float inv_s = ldexpf(1.f, 127 - e8);
      for (int e = 0; e < 32; ++e) row[t * 64 + g * 16 + e / 2] |= (uint8_t)(ToE2M1(c[e] * inv_s) << (4 * (e & 1)));
    }
}

void TileMxScales(const uint8_t* natural, int n_pad, int nsteps, bool weights_are_operand_a, uint8_t* tiled) {
  const int nblk = nsteps / 4;
  for (int t = 0; t < n_pad / kBN; ++t)
    for (int rho = 0; rho < kBN; ++rho) {
      // LDS row rho of a weight tile holds weight row n0 + perm(rho) (kernels.hip, swap_fields): the fragment rows of
      // the "weights as MFMA A operand" orientation are permuted inside each 64-row half
      int prm = rho;
      if (weights_are_operand_a) {
        const int p = (rho >> 4) & 3, g = (rho >> 2) & 3, r = rho & 3;
        prm = (rho & 64) | ((p >> 1) << 5) | (g << 3) | ((p & 1) << 2) | r;
      }
      const uint8_t* src = natural + (size_t)(t * kBN + prm) * nsteps;
      const int h = rho >> 6, w = (rho >> 4) & 3, i = rho & 15;
      for (int b = 0; b < nblk; ++b)
        for (int g = 0; g < 4; ++g) tiled[((size_t)t * nblk + b) * 512 + h * 256 + (i * 4 + g) * 4 + w] = src[4 * b + g];
    }
}

std::vector<uint8_t> PackModel(const TdnnProgram& prog, int precision) {
  if (precision < kPrecBf16x3 || precision > kPrecFp16Mx2) throw EngineError("unknown precision mode");
  const bool split = PrecWPlanes(precision) == 2;
  const bool f16 = PrecF16(precision);
  const int nl = (int)prog.layers.size();
  std::vector<BlobLayer> bl(nl);
  uint64_t cur = 0;
  for (int i = 0; i < nl; ++i) {
    const AffineLayer& L = prog.layers[i];
    BlobLayer& b = bl[i];
    memset(&b, 0, sizeof b);
    snprintf(b.name, sizeof b.name, "%s", L.out_node.c_str());
    if ((int)L.src.size() > kMaxSeg) throw EngineError("layer " + L.name + " has more than 8 Append() terms");
    b.nsrc = (int)L.src.size();
    int kp = 0;
    for (int j = 0; j < b.nsrc; ++j) {
      b.src_layer[j] = L.src[j].layer;
      b.src_offset[j] = L.src[j].offset;
      b.src_dim[j] = L.src[j].dim;
      kp += SrcKPad(L.src[j].dim, L.src[j].layer, L.segment_level, precision);
    }
    b.in_dim = L.in_dim;
    b.out_dim = L.out_dim;
    b.k_pad = kp;
    b.n_pad = RoundUp(L.out_dim, kBN);
    // wide frame-level layers in whole 256-column tiles (the senone head of prepare_nnet3_xconfig.sh:53-59: 3856 -> 4096 instead
    // of 3968): tdnn_gemm_kernel_p8 can then run them, which is worth more than the 3 % of padding
    if (!L.segment_level && b.n_pad > 1024 && (b.n_pad / kBN) % 2) b.n_pad += kBN;
    b.relu = L.relu;
    b.bn = L.bn || (f16 && split);   // scaled split-fp16 weights: the epilogue's scale step undoes the scaling
    b.log_softmax = L.log_softmax;
    b.segment_level = L.segment_level;
    b.left = L.left;
    b.right = L.right;
    const uint64_t wbytes = (uint64_t)b.n_pad * b.k_pad * 2;
    b.w_hi = cur;
    cur = Align256(cur + wbytes);
    if (split) {
      b.w_lo = cur;
      cur = Align256(cur + wbytes);
    } else {
      b.w_lo = kNone;
    }
    b.bias = cur;
    cur = Align256(cur + (uint64_t)b.n_pad * 4);
    b.scale = cur;
    cur = Align256(cur + (uint64_t)b.n_pad * 4);
    b.offset = cur;
    cur = Align256(cur + (uint64_t)b.n_pad * 4);
    // kPrecFp16Mx residual plane (frame-level layers whose K walk consists of whole blocks of four steps)
    b.w4 = b.w4_scale = b.w4b = b.w4b_scale = b.w4p = b.w4p_scale = b.w4bp = b.w4bp_scale = kNone;
    b.ldw4 = b.ldw4b = 0;
    if ((precision == kPrecAuto || precision == kPrecFp16Mx || precision == kPrecFp16Mx2) && !L.segment_level) {
      long key[kMaxSeg];
      int shift[kMaxSeg], ksteps[kMaxSeg];
      for (int j = 0; j < b.nsrc; ++j) {
        key[j] = b.src_layer[j];
        shift[j] = b.src_offset[j];
        ksteps[j] = SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, precision) / kBK;
      }
      WalkGroup wg[kMaxSeg];
      const int ng = PlanWalkGroups(b.nsrc, key, shift, ksteps, wg);
      bool ok = false;
      const int nsteps = PlanWalkSteps(ng, wg, nullptr, 0, &ok);
      if (ok && nsteps == kp / kBK) {
        b.ldw4 = nsteps / 4 * 64;
        b.w4 = cur;
        cur = Align256(cur + (uint64_t)b.n_pad * b.ldw4);
        b.w4_scale = cur;
        cur = Align256(cur + 2 * (uint64_t)b.n_pad * (uint64_t)nsteps);   // both tile orders (TileMxScales)
        bool ok64 = false;
        if (b.n_pad % 256 == 0 && PlanWalkSteps64(ng, wg, nullptr, 0, &ok64) == nsteps && ok64) {
          // whole 256-column tiles and K groups of whole 128-column blocks: tdnn_gemm_kernel_p8 can run the layer's 1.25-pass
          // launches; its blocks are pairs of 64-column K tiles, so the residual plane is packed a second time in its order
          b.w4p = cur;
          cur = Align256(cur + (uint64_t)b.n_pad * b.ldw4);
          b.w4p_scale = cur;
          cur = Align256(cur + 2 * (uint64_t)b.n_pad * (uint64_t)nsteps);
        }
      }
      if (precision == kPrecFp16Mx2) {
        // second K walk: every source another layer's output padded to whole 128-column steps (kernels.hip,
        // gemm_mx2_applicable).  A layer that cannot run it must read the network input only (it then runs kPrecFp16x3E
        // on the two planes prep_input writes): the other layers' outputs have no fp16 residual plane in this mode
        bool lo_ok = b.w4 != kNone, input_only = true;
        for (int j = 0; j < b.nsrc; ++j) {
          if (SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, precision) % 128 || b.src_layer[j] < 0) lo_ok = false;
          if (b.src_layer[j] != kSrcInput) input_only = false;
        }
        if (lo_ok) {
          b.ldw4b = nsteps / 4 * 64;
          b.w4b = cur;
          cur = Align256(cur + (uint64_t)b.n_pad * b.ldw4b);
          b.w4b_scale = cur;
          cur = Align256(cur + 2 * (uint64_t)b.n_pad * (uint64_t)nsteps);
          // tdnn_gemm_kernel_p8's second walk: tiles of 256 columns, i.e. sources of whole multiples of 256 - and only layers
          // WITHOUT time offsets: on those the kernel beats tdnn_gemm_kernel_sk in this arithmetic, on the spliced ones it loses
          // (kernels.h kP8Mx2Built).  A property of the layer: every 1.5-pass launch of the layer runs the same kernel.
          bool lo64 = kP8Mx2Built && b.w4p != kNone;
          for (int j = 0; j < b.nsrc; ++j)
            if (SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, precision) % 256 || b.src_offset[j] != 0) lo64 = false;
          if (lo64) {
            b.w4bp = cur;
            cur = Align256(cur + (uint64_t)b.n_pad * b.k_pad * 2);
            b.w4bp_scale = cur;
            cur = Align256(cur + 2 * (uint64_t)b.n_pad * (uint64_t)nsteps);
          }
        } else if (!input_only) {
          throw EngineError("precision fp16mx2 cannot run layer " + L.name + " (every source but the network input must be another layer's output)");
        }
      }
    }
  }
  BlobHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "XVHIPBLB", 8);
  h.version = kBlobVersion;
  h.precision = precision;
  h.input_dim = prog.input_dim;
  h.n_layers = nl;
  h.pooled_layer = prog.pooled_layer;
  h.pool_dim = prog.pool_dim;
  h.pool_left = prog.pool_left;
  h.pool_right = prog.pool_right;
  h.variance_floor = prog.variance_floor;
  h.output_layer = prog.output_layer;
  h.output_dim = prog.output_dim;
  h.output_is_segment = prog.output_is_segment;
  h.left_context = prog.left_context;
  h.right_context = prog.right_context;
  h.min_frames = prog.min_frames;
  h.data_offset = Align256(sizeof(BlobHeader) + (uint64_t)nl * sizeof(BlobLayer));
  h.total_bytes = h.data_offset + cur;
  std::vector<uint8_t> blob(h.total_bytes, 0);
  memcpy(blob.data(), &h, sizeof h);
  memcpy(blob.data() + sizeof h, bl.data(), (size_t)nl * sizeof(BlobLayer));
  uint8_t* data = blob.data() + h.data_offset;
  // one worker per layer (disjoint regions of the image): packing the v2 x-vector as fp16mx2 - three images per matrix - took
  // 0.14 s on one thread, a third of the start-up of a job that then needs 12 ms of device time for its 300 utterances
  auto pack_layer = [&](int i) {
    const AffineLayer& L = prog.layers[i];
    const BlobLayer& b = bl[i];
    uint16_t* whi = (uint16_t*)(data + b.w_hi);
    uint16_t* wlo = split ? (uint16_t*)(data + b.w_lo) : nullptr;
    // Split fp16: the residual plane holds values 2^-12 times smaller than the weights, which for |w| ~ 0.03 is the
    // fp16 subnormal range (< 2^-14).  The whole matrix is therefore scaled by a power of two that puts max |w| just
    // below 2^14; the exact inverse goes into the epilogue (bias * 2^S before ReLU, BatchNorm scale * 2^-S after it:
    // power-of-two factors commute with every rounding, so the result is bit-identical to the unscaled computation).
    float wscale = 1.f;
    if (f16 && split) {
      float mx = 0.f;
      for (size_t k = 0; k < L.w.size(); ++k) mx = std::max(mx, std::fabs(L.w[k]));
      if (mx > 0.f && std::isfinite(mx)) {
        int e;
        frexpf(mx, &e);              // mx = m * 2^e, m in [0.5, 1)
        wscale = ldexpf(1.f, 14 - e);
      }
    }
    for (int n = 0; n < L.out_dim; ++n) {
      int kcol = 0, kpad = 0;
      for (int j = 0; j < b.nsrc; ++j) {
        for (int d = 0; d < b.src_dim[j]; ++d) {
          const float w = L.w[(size_t)n * L.in_dim + kcol + d] * wscale;
          const size_t o = (size_t)n * b.k_pad + kpad + d;
          if (f16) {
            whi[o] = host_f32_to_f16(w);
            if (split) wlo[o] = host_f32_to_f16(w - host_f16_to_f32(whi[o]));
          } else {
            whi[o] = host_f32_to_bf16(w);
            if (split) wlo[o] = host_f32_to_bf16(w - host_bf16_to_f32(whi[o]));
          }
        }
        kcol += b.src_dim[j];
        kpad += SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, precision);
      }
    }
    if (b.w4 != kNone) {
      // e2m1 image of the residual w * 2^S - w_hi, in the order the kernels walk the K steps (kernels.h): block b of
      // four steps, lane group g, element e  <-  weight column step_wcol[4 b + e / 8] + 8 g + e % 8
      long key[kMaxSeg];
      int shift[kMaxSeg], ksteps[kMaxSeg];
      for (int j = 0; j < b.nsrc; ++j) {
        key[j] = b.src_layer[j];
        shift[j] = b.src_offset[j];
        ksteps[j] = SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, precision) / kBK;
      }
      WalkGroup wg[kMaxSeg];
      const int ng = PlanWalkGroups(b.nsrc, key, shift, ksteps, wg);
      std::vector<int> step_wcol(b.k_pad / kBK);
      PlanWalkSteps(ng, wg, step_wcol.data(), (int)step_wcol.size(), nullptr);
      // padded column -> source column of L.w (or -1)
      std::vector<int> src_col(b.k_pad, -1);
      {
        int kcol = 0, kpad = 0;
        for (int j = 0; j < b.nsrc; ++j) {
          for (int d = 0; d < b.src_dim[j]; ++d) src_col[kpad + d] = kcol + d;
          kcol += b.src_dim[j];
          kpad += SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, precision);
        }
      }
      uint8_t* w4 = data + b.w4;
      const int nsc = b.k_pad / kBK;   // scales per row
      std::vector<uint8_t> nat((size_t)b.n_pad * nsc, 127);
      std::vector<float> res(b.k_pad);
      for (int n = 0; n < L.out_dim; ++n) {
        for (int k = 0; k < b.k_pad; ++k) {
          float r = 0.f;
          if (src_col[k] >= 0) {
            const float w = L.w[(size_t)n * L.in_dim + src_col[k]] * wscale;
            r = w - host_f16_to_f32(whi[(size_t)n * b.k_pad + k]);
          }
          res[k] = r;
        }
        PackMxRow(res.data(), b.k_pad, step_wcol.data(), w4 + (size_t)n * b.ldw4, nat.data() + (size_t)n * nsc);
      }
      TileMxScales(nat.data(), b.n_pad, nsc, true, data + b.w4_scale);
      TileMxScales(nat.data(), b.n_pad, nsc, false, data + b.w4_scale + (size_t)b.n_pad * nsc);
      if (b.w4p != kNone) {   // the same residuals in the walk order of tdnn_gemm_kernel_p8
        std::vector<int> step64(b.k_pad / kBK);
        PlanWalkSteps64(ng, wg, step64.data(), (int)step64.size(), nullptr);
        std::fill(nat.begin(), nat.end(), 127);
        for (int n = 0; n < L.out_dim; ++n) {
          for (int k = 0; k < b.k_pad; ++k) {
            float r = 0.f;
            if (src_col[k] >= 0) {
              const float w = L.w[(size_t)n * L.in_dim + src_col[k]] * wscale;
              r = w - host_f16_to_f32(whi[(size_t)n * b.k_pad + k]);
            }
            res[k] = r;
          }
          PackMxRow(res.data(), b.k_pad, step64.data(), data + b.w4p + (size_t)n * b.ldw4, nat.data() + (size_t)n * nsc);
        }
        TileMxScales(nat.data(), b.n_pad, nsc, true, data + b.w4p_scale);
        TileMxScales(nat.data(), b.n_pad, nsc, false, data + b.w4p_scale + (size_t)b.n_pad * nsc);
      }
      if (b.w4b != kNone) {   // kPrecFp16Mx2: 4-bit image of the (scaled) weights in the order of the second walk
        std::vector<int> lo_wcol(b.k_pad / 128);
        const int n_lo = PlanWalkLoSteps(ng, wg, lo_wcol.data(), (int)lo_wcol.size());
        std::fill(nat.begin(), nat.end(), 127);
        std::vector<float> wrow(b.k_pad);
        for (int n = 0; n < L.out_dim; ++n) {
          for (int k = 0; k < b.k_pad; ++k) wrow[k] = src_col[k] >= 0 ? L.w[(size_t)n * L.in_dim + src_col[k]] * wscale : 0.f;
          PackMxWeightsRow(wrow.data(), lo_wcol.data(), n_lo, data + b.w4b + (size_t)n * b.ldw4b, nat.data() + (size_t)n * nsc);
        }
        TileMxScales(nat.data(), b.n_pad, nsc, true, data + b.w4b_scale);
        TileMxScales(nat.data(), b.n_pad, nsc, false, data + b.w4b_scale + (size_t)b.n_pad * nsc);
        if (b.w4bp != kNone) {   // the same image in the order of tdnn_gemm_kernel_p8's second walk, rows of k_pad * 2 bytes
          const int n64 = PlanWalkLoSteps64(ng, wg, lo_wcol.data(), (int)lo_wcol.size());
          std::fill(nat.begin(), nat.end(), 127);
          for (int n = 0; n < L.out_dim; ++n) {
            for (int k = 0; k < b.k_pad; ++k) wrow[k] = src_col[k] >= 0 ? L.w[(size_t)n * L.in_dim + src_col[k]] * wscale : 0.f;
            PackMxWeightsRow(wrow.data(), lo_wcol.data(), n64, data + b.w4bp + (size_t)n * b.k_pad * 2, nat.data() + (size_t)n * nsc);
          }
          TileMxScales(nat.data(), b.n_pad, nsc, true, data + b.w4bp_scale);
          TileMxScales(nat.data(), b.n_pad, nsc, false, data + b.w4bp_scale + (size_t)b.n_pad * nsc);
        }
      }
    }
    float* bias = (float*)(data + b.bias);
    float* scale = (float*)(data + b.scale);
    float* offset = (float*)(data + b.offset);
    const float inv = 1.f / wscale;
    for (int n = 0; n < L.out_dim; ++n) {
      bias[n] = L.bias[n] * wscale;
      scale[n] = (L.bn ? L.bn_scale[n] : 1.f) * inv;
      offset[n] = L.bn ? L.bn_offset[n] : 0.f;
    }
    // padded output columns: bias 0, scale 0, offset 0 -> they stay exactly 0 downstream
  };
  {
    std::vector<std::thread> workers;
    std::vector<std::exception_ptr> errs((size_t)nl);
    for (int i = 1; i < nl; ++i)
      workers.emplace_back([&, i]() {
        try {
          pack_layer(i);
        } catch (...) {
          errs[(size_t)i] = std::current_exception();
        }
      });
    try {
      if (nl > 0) pack_layer(0);
    } catch (...) {
      errs[0] = std::current_exception();
    }
    for (std::thread& t : workers) t.join();
    for (const std::exception_ptr& e : errs)
      if (e) std::rethrow_exception(e);
  }
  // The fingerprint of the image: header (with the field itself still zero), layer table and every packed plane.  It is what a
  // shared calibration file names (calib_file.h): a choice of arithmetic measured on one model must never be applied to another,
  // nor to the same model packed by a library whose images differ.  It travels in the header, so a context built from a
  // broadcast image on another device knows it without seeing the bytes on the host.
  const uint64_t fp = Hash64(blob.data(), blob.size());
  int32_t halves[2] = {(int32_t)(uint32_t)fp, (int32_t)(uint32_t)(fp >> 32)};
  memcpy(blob.data() + offsetof(BlobHeader, reserved), halves, sizeof halves);
  return blob;
}

const char* PrecisionName(int precision) {
  static const char* n[] = {"bf16x3", "bf16", "fp16", "fp16x3", "fp16x2", "auto", "fp16mx", "fp16mx2", "fp16x3e", "fp16mxe"};
  return precision == kPrecDefault ? "default" : (precision >= 0 && precision <= 9) ? n[precision] : "?";
}

std::vector<uint8_t> PackModelPolicy(const TdnnProgram& prog, int precision, int* resolved) {
  int p = precision;
  std::vector<uint8_t> blob;
  if (precision == kPrecDefault) {
    p = prog.output_is_segment ? (int)kPrecFp16Mx2 : (int)kPrecFp16x3;
    if (p == kPrecFp16Mx2) {
      try {
        blob = PackModel(prog, p);
      } catch (const EngineError&) {   // a layer the second K walk cannot cover: the fp32-grade three-pass mode
        p = kPrecFp16x3;
      }
    }
  }
  if (blob.empty()) blob = PackModel(prog, p);
  if (resolved) *resolved = p;
  return blob;
}

std::vector<uint8_t> ReadBlobHead(const void* device_blob, size_t n) {
  if (n < sizeof(BlobHeader)) throw EngineError("model blob too small");
  BlobHeader h;
  if (hipMemcpy(&h, device_blob, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
    throw EngineError("cannot read the header of the device-resident model blob");
  if (memcmp(h.magic, "XVHIPBLB", 8) == 0 && h.version != kBlobVersion)
    throw EngineError("model blob (device image) has format version " + std::to_string(h.version) + ", this library reads version " +
                      std::to_string(kBlobVersion) + ": repack the model");
  if (memcmp(h.magic, "XVHIPBLB", 8) != 0 || h.total_bytes != n || h.data_offset > n || h.data_offset < sizeof h)
    throw EngineError("bad model blob header (device image)");
  std::vector<uint8_t> head((size_t)h.data_offset);
  if (hipMemcpy(head.data(), device_blob, head.size(), hipMemcpyDeviceToHost) != hipSuccess)
    throw EngineError("cannot read the layer table of the device-resident model blob");
  return head;
}

BlobInfo ParseBlobInfo(const uint8_t* blob, size_t n) {
  if (n < sizeof(BlobHeader)) throw EngineError("model blob too small");
  BlobHeader h;
  memcpy(&h, blob, sizeof h);
  if (memcmp(h.magic, "XVHIPBLB", 8) != 0) throw EngineError("bad model blob header");
  if (h.version != kBlobVersion)
    throw EngineError("model blob has format version " + std::to_string(h.version) + ", this library reads version " +
                      std::to_string(kBlobVersion) + ": repack the model");
  if (h.total_bytes != n) throw EngineError("model blob size mismatch");
  if (h.n_layers < 1 || h.n_layers > 4096 || h.data_offset > n ||
      sizeof(BlobHeader) + (uint64_t)h.n_layers * sizeof(BlobLayer) > h.data_offset)
    throw EngineError("model blob: layer table does not fit the header");
  if (h.precision < kPrecBf16x3 || h.precision > kPrecFp16Mx2) throw EngineError("model blob: unknown precision mode");
  if (h.output_layer < 0 || h.output_layer >= h.n_layers || h.pooled_layer >= h.n_layers)
    throw EngineError("model blob: layer index out of range");
  const uint64_t data_bytes = n - h.data_offset;
  BlobInfo info;
  info.fingerprint = (uint64_t)(uint32_t)h.reserved[0] | ((uint64_t)(uint32_t)h.reserved[1] << 32);
  info.precision = h.precision;
  info.input_dim = h.input_dim;
  info.pooled_layer = h.pooled_layer;
  info.pool_dim = h.pool_dim;
  info.pool_left = h.pool_left;
  info.pool_right = h.pool_right;
  info.variance_floor = h.variance_floor;
  info.output_layer = h.output_layer;
  info.output_dim = h.output_dim;
  info.output_is_segment = h.output_is_segment;
  info.left_context = h.left_context;
  info.right_context = h.right_context;
  info.min_frames = h.min_frames;
  for (int i = 0; i < h.n_layers; ++i) {
    BlobLayer b;
    memcpy(&b, blob + sizeof h + (size_t)i * sizeof(BlobLayer), sizeof b);
    {
      // everything the engine later turns into device pointers or loop bounds
      auto inside = [&](uint64_t off, uint64_t bytes) { return off <= data_bytes && bytes <= data_bytes - off; };
      bool ok = b.nsrc >= 1 && b.nsrc <= kMaxSeg && b.n_pad > 0 && b.n_pad % kBN == 0 && b.k_pad > 0 && b.k_pad % kBK == 0 &&
                b.out_dim >= 1 && b.out_dim <= b.n_pad && b.in_dim >= 1;
      int kp = 0;
      for (int j = 0; ok && j < b.nsrc; ++j) {
        ok = b.src_layer[j] >= kSrcPooled && b.src_layer[j] < i && b.src_dim[j] >= 1 && std::abs(b.src_offset[j]) <= 15;
        kp += SrcKPad(b.src_dim[j], b.src_layer[j], b.segment_level, h.precision);
      }
      ok = ok && kp == b.k_pad;
      const uint64_t wbytes = (uint64_t)b.n_pad * b.k_pad * 2;
      ok = ok && inside(b.w_hi, wbytes) && (b.w_lo == kNone || inside(b.w_lo, wbytes)) && inside(b.bias, (uint64_t)b.n_pad * 4) &&
           inside(b.scale, (uint64_t)b.n_pad * 4) && inside(b.offset, (uint64_t)b.n_pad * 4);
      if (b.w4 != kNone)
        ok = ok && b.ldw4 == b.k_pad / kBK / 4 * 64 && (b.k_pad / kBK) % 4 == 0 && inside(b.w4, (uint64_t)b.n_pad * b.ldw4) &&
             b.w4_scale != kNone && inside(b.w4_scale, 2 * (uint64_t)b.n_pad * (uint64_t)(b.k_pad / kBK));
      if (b.w4b != kNone)
        ok = ok && b.w4 != kNone && b.ldw4b == b.k_pad / 128 * 64 && inside(b.w4b, (uint64_t)b.n_pad * b.ldw4b) &&
             b.w4b_scale != kNone && inside(b.w4b_scale, 2 * (uint64_t)b.n_pad * (uint64_t)(b.k_pad / kBK));
      // the same images in the K-walk order of tdnn_gemm_kernel_p8 (these become device pointers the kernel DMAs
      // n_pad * ldw4 / n_pad * k_pad * 2 bytes from, a GPU fault is fatal): each needs what it is a re-ordering of, its scales,
      // and whole 256-column tiles
      const uint64_t sc_bytes = 2 * (uint64_t)b.n_pad * (uint64_t)(b.k_pad / kBK);
      if (b.w4p != kNone)
        ok = ok && b.w4 != kNone && b.n_pad % 256 == 0 && inside(b.w4p, (uint64_t)b.n_pad * b.ldw4) && b.w4p_scale != kNone &&
             inside(b.w4p_scale, sc_bytes);
      else
        ok = ok && b.w4p_scale == kNone;
      if (b.w4bp != kNone)
        ok = ok && b.w4p != kNone && b.w4b != kNone && inside(b.w4bp, (uint64_t)b.n_pad * b.k_pad * 2) && b.w4bp_scale != kNone &&
             inside(b.w4bp_scale, sc_bytes);
      else
        ok = ok && b.w4bp_scale == kNone;
      if (!ok) throw EngineError("model blob: layer " + std::to_string(i) + " is inconsistent");
    }
    BlobLayerInfo li;
    li.name = b.name;
    li.in_dim = b.in_dim;
    li.out_dim = b.out_dim;
    li.k_pad = b.k_pad;
    li.n_pad = b.n_pad;
    li.relu = b.relu;
    li.bn = b.bn;
    li.log_softmax = b.log_softmax;
    li.segment_level = b.segment_level;
    li.left = b.left;
    li.right = b.right;
    li.has_w4 = b.w4 != kNone;
    li.has_w4b = b.w4b != kNone;
    li.has_w4p = b.w4p != kNone;
    for (int j = 0; j < b.nsrc; ++j) {
      LayerSource s;
      s.layer = b.src_layer[j];
      s.offset = b.src_offset[j];
      s.dim = b.src_dim[j];
      li.src.push_back(s);
    }
    info.layers.push_back(li);
  }
  return info;
}

double BlobInfo::Macs(int T) const {
  const int n = (int)layers.size();
  std::vector<int> nl(n, 1 << 30), nr(n, 1 << 30);
  const int root = output_is_segment ? pooled_layer : output_layer;
  nl[root] = layers[root].left;
  nr[root] = layers[root].right;
  for (int i = n - 1; i >= 0; --i) {
    if (layers[i].segment_level || nl[i] == (1 << 30)) continue;
    for (const LayerSource& s : layers[i].src)
      if (s.layer >= 0) {
        nl[s.layer] = std::min(nl[s.layer], nl[i] + s.offset);
        nr[s.layer] = std::min(nr[s.layer], nr[i] - s.offset);
      }
  }
  double macs = 0;
  for (int i = 0; i < n; ++i) {
    if (layers[i].segment_level) macs += (double)layers[i].in_dim * layers[i].out_dim;
    else if (nl[i] != (1 << 30)) macs += (double)layers[i].in_dim * layers[i].out_dim * std::max(0, T - nl[i] - nr[i]);
  }
  return macs;
}

// ------------------------------------------------------------------------------------------- Engine
void Engine::Check(hipError_t e, const char* what) const {
  if (e != hipSuccess) {
    std::ostringstream m;
    m << what << ": " << hipGetErrorString(e) << " (HIP error " << (int)e << ", device " << device_ << ")";
    throw EngineError(m.str());
  }
}

Engine::Engine(const uint8_t* blob, size_t n, int device, const void* device_image) : device_(device) {
  info_ = ParseBlobInfo(blob, n);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    throw EngineError("no HIP device available: this library has no CPU path (hipGetDeviceCount: " +
                      std::string(hipGetErrorString(e)) + ")");
  if (device < 0 || device >= count) throw EngineError("HIP device index out of range");
  Check(hipSetDevice(device_), "hipSetDevice");
  hipDeviceProp_t prop;
  Check(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    throw EngineError(std::string("kernels are built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
  frame_mode_ = !info_.output_is_segment;
  if (frame_mode_) {
    // nnet3-compute semantics: every input frame gets an output row; the chunk is extended by edge replication
    if (info_.pooled_layer >= 0) throw EngineError("a frame-level output that depends on statistics pooling is not supported");
    pad_left_ = info_.left_context;
    pad_right_ = info_.right_context;
  } else {
    if (info_.pooled_layer < 0) throw EngineError("model has no statistics pooling");
    for (size_t i = 0; i < info_.layers.size(); ++i)
      for (const LayerSource& s : info_.layers[i].src)
        if (s.layer == info_.pooled_layer)
          throw EngineError("the pooled layer must feed only the statistics pooling");
  }
  nplanes_ = PrecWPlanes(info_.precision);   // residual planes exist for every split mode
  // Kernel modes.  slow_prec_ runs everything that is not a frame-level GEMM of a "fast" chunk; fast chunks (only
  // kPrecFp16x2 / kPrecAuto have them) are those that pool at least fast_min_pooled_ frames, see FillPlan.
  const bool fast_family = info_.precision == kPrecFp16x2 || info_.precision == kPrecAuto || info_.precision == kPrecFp16Mx ||
                           info_.precision == kPrecFp16Mx2;
  slow_prec_ = fast_family ? (int)kPrecFp16x3 : info_.precision;
  has_fast_ = !frame_mode_ && fast_family;
  // fast chunks run kPrecFp16Mx on the layers that allow it (packed residual plane, sources with a group-max table)
  // and kPrecFp16x2 on the others
  fast_mx_ = has_fast_ && (info_.precision == kPrecAuto || info_.precision == kPrecFp16Mx || info_.precision == kPrecFp16Mx2);
  fast_mx2_ = has_fast_ && info_.precision == kPrecFp16Mx2;
  fast_min_pooled_ = 0;
  if (info_.precision == kPrecAuto || info_.precision == kPrecFp16Mx2) {
    // kPrecFp16Mx2 corrects the activation rounding, but what is left still averages over the pooled frames: 3-4e-5 at
    // 386 pooled frames, 4.5e-5 at 123, 0.6-1.5e-4 at 11 - chunks below the threshold take the three-pass arithmetic
    const char* e = getenv("XVEC_FAST_MIN_POOLED");
    fast_min_pooled_ = (e && *e) ? atoi(e) : (info_.precision == kPrecAuto ? kDefaultFastMinPooled : kDefaultFastMinPooledMx2);
    mx2_min_pooled_ = (e && *e) ? atoi(e) : kDefaultFastMinPooledMx2;
    mx_min_pooled_ = (e && *e) ? atoi(e) : kDefaultFastMinPooled;
  }
  fast_mode_ = info_.precision;
  Check(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
  {
    const char* e = getenv("XVEC_LANES");
    int nl = e && *e ? atoi(e) : 2;
    if (nl < 1) nl = 1;
    if (nl > kMaxLanes) nl = kMaxLanes;   // (CheckKernelFaults keeps one bit per lane below the engine-stream and caller-stream bits)
    lanes_.resize(nl);
    for (Lane& L : lanes_) {
      Check(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking), "hipStreamCreate(lane)");
      Check(hipEventCreateWithFlags(&L.done, hipEventDisableTiming), "hipEventCreate(lane)");
      L.act.resize(info_.layers.size());
    }
  }

  BlobHeader h;
  memcpy(&h, blob, sizeof h);
  blob_data_bytes_ = (size_t)(h.total_bytes - h.data_offset);
  Check(hipMalloc(&d_blob_, blob_data_bytes_), "hipMalloc(weights)");
  if (device_image)   // the image is already on this device (e.g. it arrived by RCCL broadcast): no host round trip
    Check(hipMemcpy(d_blob_, (const uint8_t*)device_image + h.data_offset, blob_data_bytes_, hipMemcpyDeviceToDevice),
          "hipMemcpy(weights, device to device)");
  else
    Check(hipMemcpy(d_blob_, blob + h.data_offset, blob_data_bytes_, hipMemcpyHostToDevice), "hipMemcpy(weights)");
  layers_.resize(info_.layers.size());
  for (size_t i = 0; i < layers_.size(); ++i) {
    BlobLayer b;
    memcpy(&b, blob + sizeof h + i * sizeof(BlobLayer), sizeof b);
    const uint8_t* base = (const uint8_t*)d_blob_;
    layers_[i].w_hi = (const uint16_t*)(base + b.w_hi);
    layers_[i].w_lo = b.w_lo == kNone ? nullptr : (const uint16_t*)(base + b.w_lo);
    layers_[i].bias = (const float*)(base + b.bias);
    layers_[i].scale = (const float*)(base + b.scale);
    layers_[i].offset = (const float*)(base + b.offset);
    layers_[i].w4 = b.w4 == kNone ? nullptr : base + b.w4;
    layers_[i].w4_scale = b.w4_scale == kNone ? nullptr : base + b.w4_scale;
    layers_[i].ldw4 = b.ldw4;
    layers_[i].w4b = b.w4b == kNone ? nullptr : base + b.w4b;
    layers_[i].w4b_scale = b.w4b_scale == kNone ? nullptr : base + b.w4b_scale;
    layers_[i].ldw4b = b.ldw4b;
    layers_[i].w4p = b.w4p == kNone ? nullptr : base + b.w4p;
    layers_[i].w4p_scale = b.w4p_scale == kNone ? nullptr : base + b.w4p_scale;
    layers_[i].w4bp = b.w4bp == kNone ? nullptr : base + b.w4bp;
    layers_[i].w4bp_scale = b.w4bp_scale == kNone ? nullptr : base + b.w4bp_scale;
  }
  // the weight upload above ran on the null stream, which the engine's non-blocking streams are not ordered behind (and a
  // device-to-device copy returns before it has run): it is over before the constructor returns.  Waiting for the null stream
  // is enough - other contexts' (non-blocking) streams on this device keep running
  Check(hipStreamSynchronize(nullptr), "hipStreamSynchronize(weights)");
  {
    use_p8_ = DebugKnobInt("p8", 1) != 0;   // 0: never run tdnn_gemm_kernel_p8 (A/B against the 32-column kernels)
    // XVEC_DEBUG=p8_whole=... (read per context): how tdnn_gemm_kernel_p8 deals its work out.  0 (default): K tiles evenly, partial tiles
    // exchanged through the workspace - the shortest launch when a launch has the GPU to itself.  1: whole output tiles only -
    // every launch is LONGER on its own (800 tiles on 256 workgroups: a quarter of them gets a fourth tile) but needs no exchange,
    // and with two batches in flight the other lane's kernels fill the CUs that finish early: +3 % throughput at two lanes,
    // -10 % on the dominant kernel's own roofline fraction (profiles/r05_p8_whole_tiles.md).  2: whole tiles for the K <= 512 layers.
    // Same bits either way: every output element is accumulated in one fixed order.
    p8_whole_ = DebugKnobInt("p8_whole", 0);
  }
  in_ld_ = RoundUp(info_.input_dim, kBK);
  stats_ld_ = RoundUp(2 * info_.pool_dim, kBK);
  // Layers whose every source is the network input run tdnn_first_kernel (kernels.h) in the split-precision families:
  // fp32 features in, planes out, weights resident on the CU; prep_input then only runs for whatever else reads the input.
  {
    const bool allow = DebugKnobInt("first_kernel", 1) != 0 && nplanes_ == 2;
    need_prep_ = false;
    for (size_t i = 0; i < layers_.size(); ++i) {
      const BlobLayerInfo& li = info_.layers[i];
      bool reads_input = false, only_input = !li.segment_level && !li.src.empty();
      int off[kMaxSeg] = {0};
      for (size_t j = 0; j < li.src.size(); ++j) {
        if (li.src[j].layer == kSrcInput) reads_input = true;
        if (li.src[j].layer != kSrcInput || li.src[j].dim != info_.input_dim) only_input = false;
        off[j] = li.src[j].offset;
      }
      const bool planes_out = (int)i != info_.pooled_layer && !(frame_mode_ && (int)i == info_.output_layer);
      if (allow && only_input && planes_out && layers_[i].w_lo && FirstLayerApplicable(info_.input_dim, (int)li.src.size(), off)) {
        first_buf_.emplace_back();
        first_buf_.emplace_back();
        Buf& bh = first_buf_[first_buf_.size() - 2];
        Buf& bl = first_buf_[first_buf_.size() - 1];
        Ensure(&bh, (size_t)li.n_pad * kFirstK * 2, false);
        Ensure(&bl, (size_t)li.n_pad * kFirstK * 2, false);
        const int seg_pad = SrcKPad(info_.input_dim, kSrcInput, false, info_.precision);
        Check(launch_compact_first(layers_[i].w_hi, li.k_pad, seg_pad, li.n_pad, (int)li.src.size(), info_.input_dim, (uint16_t*)bh.p, stream_),
              "compact_first launch");
        Check(launch_compact_first(layers_[i].w_lo, li.k_pad, seg_pad, li.n_pad, (int)li.src.size(), info_.input_dim, (uint16_t*)bl.p, stream_),
              "compact_first launch");
        layers_[i].first = true;
        layers_[i].wc_hi = (const uint16_t*)bh.p;
        layers_[i].wc_lo = (const uint16_t*)bl.p;
      } else if (reads_input) {
        need_prep_ = true;
      }
    }
    Check(hipStreamSynchronize(stream_), "hipStreamSynchronize(compact_first)");
  }
}

Engine::~Engine() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  plan_cache_.clear();
  auto fr = [](Buf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
  };
  if (stream_) release_stream_workspace(stream_);
  for (Lane& L : lanes_) {
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    if (L.stream) release_stream_workspace(L.stream);
    fr(L.gmax);
    for (ActBuf& a : L.act) {
      fr(a.act_hi);
      fr(a.act_lo);
      fr(a.act_lo4);
      fr(a.act_lo4s);
    }
    fr(L.in_hi);
    fr(L.in_lo);
    fr(L.partial);
    fr(L.stats_hi);
    fr(L.stats_lo);
    fr(L.out_f32);
    fr(L.splitk_ws);
    fr(L.frame_f32);
    if (L.done) (void)hipEventDestroy(L.done);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  for (HostSlot& S : host_slots_) {
    if (S.done) {
      (void)hipEventSynchronize(S.done);
      (void)hipEventDestroy(S.done);
    }
    if (S.h2d_done) (void)hipEventDestroy(S.h2d_done);
    S.plan.reset();
    fr(S.d_feats);
    fr(S.d_out);
    fr(S.d_tables);
    fr(S.d_raw);
    fr(S.d_prefix);
    fr(S.d_fetab);
    fr(S.d_cm);
    if (S.h_fetab) (void)hipHostFree(S.h_fetab);
    if (S.h_feats) (void)hipHostFree(S.h_feats);
    if (S.h_out) (void)hipHostFree(S.h_out);
    if (S.h_tables) (void)hipHostFree(S.h_tables);
  }
  for (Buf& b : first_buf_) fr(b);
  fr(feats_stage_);
  fr(fe_raw_);
  fr(fe_tab_);
  fr(fe_prefix_);
  fr(fe_out_);
  fr(out_stage_);
  if (d_blob_) (void)hipFree(d_blob_);
  if (copy_stream_) (void)hipStreamDestroy(copy_stream_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::Ensure(Buf* b, size_t bytes, bool zero, hipStream_t consumer) {
  if (b->bytes >= bytes && b->p) return;
  if (b->p) Check(hipFree(b->p), "hipFree");
  b->p = nullptr;
  Check(hipMalloc(&b->p, bytes), "hipMalloc");
  b->bytes = bytes;
  if (zero) {
    // The fill is ordered on the stream that will consume the buffer (hipMemsetAsync on `consumer`): a plain hipMemset runs on
    // the null stream, which the engine's non-blocking streams are not ordered behind - with another stream keeping the chip
    // busy the FIRST forward pass of a context could run beside the zero-fill of its own activation planes (round 4,
    // tools/stress_frames.py).  Round 4 waited for the whole device here; only the ordering against one stream is needed
    // (four processes / threads per GPU is the recipes' launch mode, and a device-wide wait stalls them all).
    // Later users of the buffer on another stream are ordered behind this one by the lane's `done` event.
    Check(hipMemsetAsync(b->p, 0, bytes, consumer), "hipMemsetAsync");
  }
}

void Engine::EnsureCapacity(Lane& L, int rows, int b_pad, hipStream_t s) {
  // Growing = freeing planes this lane's previous batch may still be using: wait for THAT batch (hipFree then does what it
  // does); the fills of the new planes are ordered on `s`, the stream about to consume them.  No device-wide wait of ours.
  bool grew = false;
  if (rows > L.cap_rows) {
    grew = true;
    if (L.busy) Check(hipEventSynchronize(L.done), "hipEventSynchronize(lane)");
    const size_t r = (size_t)rows + 2 * kHalo;
    Ensure(&L.in_hi, r * in_ld_ * 2, true, s);
    if (nplanes_ == 2) Ensure(&L.in_lo, r * in_ld_ * 2, true, s);
    for (size_t i = 0; i < layers_.size(); ++i) {
      const BlobLayerInfo& li = info_.layers[i];
      if (li.segment_level) continue;
      if ((int)i == info_.pooled_layer) {
        Ensure(&L.partial, (size_t)(rows / kRowAlign) * 2 * li.n_pad * 4, true, s);
        continue;
      }
      if (frame_mode_ && (int)i == info_.output_layer) {
        Ensure(&L.frame_f32, (size_t)rows * li.n_pad * (logits16() ? 2 : 4), false);
        continue;
      }
      Ensure(&L.act[i].act_hi, r * li.n_pad * 2, true, s);
      if (nplanes_ == 2) Ensure(&L.act[i].act_lo, r * li.n_pad * 2, true, s);
      if (info_.precision == kPrecFp16Mx2 && !frame_mode_) {   // whatever the current fast mode is (SetFastMode)
        Ensure(&L.act[i].act_lo4, r * (li.n_pad / 2), true, s);
        Ensure(&L.act[i].act_lo4s, r * Lo4ScalePitch(li.n_pad), true, s);
      }
    }
    // per layer: max |activation| of every 16-row group (what a kPrecFp16Mx consumer scales its 4-bit copy by)
    if (fast_mx_ || can_switch_fast_mode()) Ensure(&L.gmax, layers_.size() * (size_t)(rows / kRowAlign) * 4, true, s);
    L.gmax_stride = rows / kRowAlign;
    L.cap_rows = rows;
  }
  if (!frame_mode_ && b_pad > L.cap_b) {
    if (L.busy) Check(hipEventSynchronize(L.done), "hipEventSynchronize(lane)");
    Ensure(&L.stats_hi, (size_t)b_pad * stats_ld_ * 2, true, s);
    if (nplanes_ == 2) Ensure(&L.stats_lo, (size_t)b_pad * stats_ld_ * 2, true, s);
    for (size_t i = 0; i < layers_.size(); ++i) {
      const BlobLayerInfo& li = info_.layers[i];
      if (!li.segment_level) continue;
      if ((int)i == info_.output_layer) {
        Ensure(&L.out_f32, (size_t)b_pad * li.n_pad * 4, true, s);
      } else {
        Ensure(&L.act[i].act_hi, (size_t)b_pad * li.n_pad * 2, true, s);
        if (nplanes_ == 2) Ensure(&L.act[i].act_lo, (size_t)b_pad * li.n_pad * 2, true, s);
      }
    }
    L.cap_b = b_pad;
    grew = true;
  }
  // The fills above are ordered on `s`, which is all the consuming kernels need.  They are nevertheless waited for here, once per
  // growth (the first batches of a job): the planes' halo rows, the group maxima and the partial sums are READ by kernels of
  // every later pass without ever being written again, and a fill that is still queued when, say, the other lane's first batch
  // frees and reallocates its own planes is an ordering this code would rather not reason about (VERDICT / ADVICE r05 suspect
  // (a) of the co-tenancy failure; not reproduced, removed anyway).  One stream, not the device: other contexts are not stalled.
  if (grew) Check(hipStreamSynchronize(s), "hipStreamSynchronize(fills)");
}

uint16_t* Engine::ActBase(const Buf& b, int ld) const {
  return b.p ? (uint16_t*)b.p + (size_t)kHalo * ld : nullptr;
}

Engine::Plan::~Plan() {
  if (d_tables && !borrowed_tables) (void)hipFree(d_tables);
}

void Engine::FillPlan(const int32_t* row_offsets, int B, Plan* plan, std::vector<uint8_t>* tables) const {
  if (B <= 0) throw EngineError("empty batch");
  std::vector<int32_t> key(B);
  for (int b = 0; b < B; ++b) {
    key[b] = row_offsets[b + 1] - row_offsets[b];
    if (frame_mode_ ? key[b] < 1 : key[b] < info_.min_frames) {
      std::ostringstream m;
      m << "chunk " << b << " has " << key[b] << " frames; the network needs at least " << info_.min_frames;
      throw EngineError(m.str());
    }
  }
  plan->B = B;
  plan->b_pad = RoundUp(B, kBM);
  plan->src_off.assign(row_offsets, row_offsets + B + 1);
  plan->src_rows = row_offsets[B];
  // Device row layout: the chunks that take the two-pass kernels ("fast": they pool enough frames for the
  // activation rounding error to average out) come first, then, from a 256-row boundary on, the others.  Which
  // region a chunk lands in depends on its own length only, so its embedding does not depend on the batch around it.
  const BlobLayerInfo& pl = info_.layers[frame_mode_ ? 0 : info_.pooled_layer];
  const int pool_first = std::max(pl.left, -info_.pool_left);
  std::vector<int32_t> dev_off(B), cnt(B, 0);
  std::vector<char> fast(B, 0);
  for (int b = 0; b < B && !frame_mode_; ++b) {
    // frames of the pooled layer that exist, intersected with the pooling window of output index t=0
    const int last = std::min(key[b] - 1 - pl.right, info_.pool_right);
    if (last < pool_first) throw EngineError("chunk has no frame inside the pooling window");
    cnt[b] = last - pool_first + 1;
    fast[b] = has_fast_ && cnt[b] >= fast_min_pooled_;
  }
  long off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int b = 0; b < B; ++b) {
      if ((pass == 0) != (fast[b] != 0)) continue;
      dev_off[b] = (int32_t)off;
      off += RoundUp(key[b] + pad_left_ + pad_right_, kRowAlign);
      if (off > (1l << 30)) throw EngineError("batch too large (more than 2^30 device rows)");
    }
    // the 256-row GEMM variant needs an even number of 128-row tiles; the region of the two-pass kernels is cut into
    // 512-row tiles by the stream-K variant
    off = RoundUp((int)off, (pass == 0 && has_fast_) ? 4 * kBM : 2 * kBM);
    if (pass == 0) plan->rows_fast = (int)off;
  }
  plan->rows = (int)off;
  const int ngrp = plan->rows / kRowAlign;
  std::vector<int32_t> grp_utt(ngrp, -1), g0(B), g1(B);
  std::vector<int8_t> grp_range((size_t)ngrp * 2, 0);
  std::vector<int32_t> out_row;
  if (frame_mode_) {
    plan->out_off.assign(1, 0);
    for (int b = 0; b < B; ++b) {
      const int Td = key[b] + pad_left_ + pad_right_;
      const int ga = dev_off[b] / kRowAlign, gb = (dev_off[b] + RoundUp(Td, kRowAlign)) / kRowAlign;
      for (int g = ga; g < gb; ++g) grp_utt[g] = b;
      // output frame i of the chunk is frame pad_left + i of the padded chunk
      for (int i = 0; i < key[b]; ++i) out_row.push_back(dev_off[b] + pad_left_ + i);
      plan->out_off.push_back((int32_t)out_row.size());
    }
    plan->n_out = (int)out_row.size();
  }
  for (int b = 0; b < B && !frame_mode_; ++b) {
    const int T = key[b];
    const int first = pool_first;
    const int last = first + cnt[b] - 1;
    g0[b] = dev_off[b] / kRowAlign;
    g1[b] = (dev_off[b] + RoundUp(T, kRowAlign)) / kRowAlign;
    for (int g = g0[b]; g < g1[b]; ++g) {
      grp_utt[g] = b;
      const int t0 = (g - g0[b]) * kRowAlign;
      grp_range[2 * g] = (int8_t)std::min(std::max(first - t0, 0), kRowAlign);
      grp_range[2 * g + 1] = (int8_t)std::min(std::max(last + 1 - t0, 0), kRowAlign);
    }
  }
  // per layer: the rows of every 16-row group that are computable frames of that layer (kPrecFp16Mx group maxima)
  const int nlay = (int)info_.layers.size();
  std::vector<int8_t> act_range;
  if (fast_mx_ && !frame_mode_) {
    act_range.assign((size_t)nlay * ngrp * 2, 0);
    for (int i = 0; i < nlay; ++i) {
      const BlobLayerInfo& li = info_.layers[i];
      if (li.segment_level) continue;
      int8_t* t = act_range.data() + (size_t)i * ngrp * 2;
      for (int b = 0; b < B; ++b) {
        const int lo = li.left, hi = key[b] - li.right;   // computable frames [lo, hi)
        for (int g = g0[b]; g < g1[b]; ++g) {
          const int t0 = (g - g0[b]) * kRowAlign;
          t[2 * g] = (int8_t)std::min(std::max(lo - t0, 0), kRowAlign);
          t[2 * g + 1] = (int8_t)std::min(std::max(hi - t0, 0), kRowAlign);
        }
      }
    }
  }
  plan->ngrp = ngrp;
  // one table image, 256-B aligned sections
  plan->o_src = 0;
  plan->o_dev = Align256(plan->o_src + (size_t)(B + 1) * 4);
  plan->o_gu = Align256(plan->o_dev + (size_t)B * 4);
  plan->o_gr = Align256(plan->o_gu + (size_t)ngrp * 4);
  plan->o_g0 = Align256(plan->o_gr + (size_t)ngrp * 2);
  plan->o_g1 = Align256(plan->o_g0 + (size_t)B * 4);
  plan->o_cn = Align256(plan->o_g1 + (size_t)B * 4);
  plan->o_or = Align256(plan->o_cn + (size_t)B * 4);
  plan->o_ar = Align256(plan->o_or + out_row.size() * 4);
  plan->o_gs = Align256(plan->o_ar + act_range.size());
  // source of every 16-row group for the first-layer kernel (kernels.h, FirstArgs::grp_src)
  std::vector<int32_t> grp_src((size_t)ngrp * 4, 0);
  for (int g = 0; g < ngrp; ++g) {
    const int b = grp_utt[g];
    if (b < 0) continue;
    const int t0 = g * kRowAlign - dev_off[b];
    const int s0 = row_offsets[b], len = key[b];
    grp_src[4 * g] = s0 + t0 - pad_left_;
    grp_src[4 * g + 1] = len + pad_left_ + pad_right_ - t0;
    grp_src[4 * g + 2] = s0;
    grp_src[4 * g + 3] = s0 + len - 1;
  }
  const size_t total = Align256(plan->o_gs + grp_src.size() * 4);
  std::vector<uint8_t>& host = *tables;
  host.assign(total, 0);
  if (!act_range.empty()) memcpy(host.data() + plan->o_ar, act_range.data(), act_range.size());
  memcpy(host.data() + plan->o_gs, grp_src.data(), grp_src.size() * 4);
  if (!out_row.empty()) memcpy(host.data() + plan->o_or, out_row.data(), out_row.size() * 4);
  memcpy(host.data() + plan->o_src, plan->src_off.data(), (size_t)(B + 1) * 4);
  memcpy(host.data() + plan->o_dev, dev_off.data(), (size_t)B * 4);
  memcpy(host.data() + plan->o_gu, grp_utt.data(), (size_t)ngrp * 4);
  memcpy(host.data() + plan->o_gr, grp_range.data(), (size_t)ngrp * 2);
  memcpy(host.data() + plan->o_g0, g0.data(), (size_t)B * 4);
  memcpy(host.data() + plan->o_g1, g1.data(), (size_t)B * 4);
  memcpy(host.data() + plan->o_cn, cnt.data(), (size_t)B * 4);
}

void Engine::BindPlan(Plan* plan, const void* device_tables) {
  const uint8_t* d = (const uint8_t*)device_tables;
  plan->d_src_off = (const int32_t*)(d + plan->o_src);
  plan->d_dev_off = (const int32_t*)(d + plan->o_dev);
  plan->d_grp_utt = (const int32_t*)(d + plan->o_gu);
  plan->d_grp_range = (const int8_t*)(d + plan->o_gr);
  plan->d_utt_grp0 = (const int32_t*)(d + plan->o_g0);
  plan->d_utt_grp1 = (const int32_t*)(d + plan->o_g1);
  plan->d_utt_count = (const int32_t*)(d + plan->o_cn);
  plan->d_out_row = (const int32_t*)(d + plan->o_or);
  plan->d_act_range = (const int8_t*)(d + plan->o_ar);
  plan->d_grp_src = d + plan->o_gs;
}

std::shared_ptr<Engine::Plan> Engine::MakePlan(const int32_t* row_offsets, int B) {
  if (B <= 0) throw EngineError("empty batch");
  std::vector<int32_t> key(B + 1);
  for (int b = 0; b < B; ++b) key[b] = row_offsets[b + 1] - row_offsets[b];
  key[B] = row_offsets[0];
  auto hit = plan_cache_.find(key);
  if (hit != plan_cache_.end()) return hit->second;
  Check(hipSetDevice(device_), "hipSetDevice");
  auto plan = std::make_shared<Plan>();
  std::vector<uint8_t> host;
  FillPlan(row_offsets, B, plan.get(), &host);
  Check(hipMalloc(&plan->d_tables, host.size()), "hipMalloc(plan)");
  Check(hipMemcpy(plan->d_tables, host.data(), host.size(), hipMemcpyHostToDevice), "hipMemcpy(plan)");
  Check(hipStreamSynchronize(nullptr), "hipStreamSynchronize(plan)");   // null-stream copy; the kernels run on non-blocking streams
  BindPlan(plan.get(), plan->d_tables);
  if (plan_cache_.size() >= 64) plan_cache_.clear();
  plan_cache_[key] = plan;
  return plan;
}

void Engine::Forward(const Plan& plan, const float* feats_dev, float* out_dev, int out_ld, hipStream_t stream) {
  // lane selection: the engine's own streams rotate; a caller-provided stream is mapped to a lane and ordered
  // behind whatever that lane did last (its buffers are reused)
  ForwardOnLane(stream ? ((size_t)stream >> 6) % lanes_.size() : (next_lane_++ % lanes_.size()), plan, feats_dev, out_dev,
                out_ld, stream);
}

void Engine::ForwardOnLane(size_t lane, const Plan& plan, const float* feats_dev, float* out_dev, int out_ld,
                           hipStream_t stream) {
  Check(hipSetDevice(device_), "hipSetDevice");
  Lane& L = lanes_[lane % lanes_.size()];
  hipStream_t s = stream ? stream : L.stream;
  if (stream && stream != stream_ && std::find(ext_streams_.begin(), ext_streams_.end(), stream) == ext_streams_.end()) {
    bool own = false;
    for (const Lane& l : lanes_) own = own || l.stream == stream;
    if (!own) ext_streams_.push_back(stream);
  }
  if (L.busy) Check(hipStreamWaitEvent(s, L.done, 0), "hipStreamWaitEvent(lane)");
  EnsureCapacity(L, plan.rows, plan.b_pad, s);
  const int prec = slow_prec_;

  PrepArgs pa;
  pa.feats = feats_dev;
  pa.src_off = plan.d_src_off;
  pa.dev_off = plan.d_dev_off;
  pa.grp_utt = plan.d_grp_utt;
  pa.rows = plan.rows;
  pa.dim = info_.input_dim;
  pa.ld = in_ld_;
  pa.out_hi = ActBase(L.in_hi, in_ld_);
  pa.out_lo = ActBase(L.in_lo, in_ld_);
  pa.pad_left = pad_left_;
  pa.pad_right = pad_right_;
  const bool mx_pass = fast_mx_ && plan.rows_fast > 0;
  pa.zero_words = mx_pass ? (unsigned*)L.gmax.p : nullptr;
  pa.n_zero_words = mx_pass ? (int)(layers_.size() * (size_t)L.gmax_stride) : 0;
  auto gmax_of = [&](int layer) { return (unsigned*)L.gmax.p + (size_t)layer * L.gmax_stride; };
  // profiling: every launch_* call below is bracketed by its own (start, stop) event pair, stamped by the dispatch
  // itself (kernels.hip: set_launch_events) - no extra packets between the kernels of the timed region
  std::vector<hipEvent_t> prof_run;
  const bool first_prof = prof_on_ && prof_labels_.empty();
  auto arm = [&](const std::string& label) {
    if (!prof_on_) return;
    hipEvent_t e0, e1;
    Check(hipEventCreate(&e0), "hipEventCreate");
    Check(hipEventCreate(&e1), "hipEventCreate");
    prof_run.push_back(e0);
    prof_run.push_back(e1);
    if (first_prof) prof_labels_.push_back(label);
    set_launch_events(e0, e1);
  };
  auto disarm = [&]() {
    if (prof_on_) set_launch_events(nullptr, nullptr);
  };
  if (need_prep_) {
    arm("prep_input");
    Check(launch_prep_input(pa, prec, s), "prep_input launch");
    disarm();
  } else if (pa.n_zero_words > 0) {
    // the group-max tables of this pass, otherwise cleared by prep_input
    Check(hipMemsetAsync(pa.zero_words, 0, (size_t)pa.n_zero_words * 4, s), "hipMemsetAsync(group maxima)");
  }

  bool direct_out = false;   // the output layer wrote into out_dev itself
  for (size_t i = 0; i < layers_.size(); ++i) {
    const BlobLayerInfo& li = info_.layers[i];
    DevLayer& dl = layers_[i];
    GemmArgs ga;
    memset(&ga, 0, sizeof ga);
    ga.nseg = (int)li.src.size();
    int ksteps = 0;
    for (int j = 0; j < ga.nseg; ++j) {
      const LayerSource& src = li.src[j];
      Seg& sg = ga.seg[j];
      if (src.layer == kSrcInput) {
        sg.hi = ActBase(L.in_hi, in_ld_);
        sg.lo = ActBase(L.in_lo, in_ld_);
        sg.ld = in_ld_;
      } else if (src.layer == kSrcPooled) {
        sg.hi = (const uint16_t*)L.stats_hi.p;
        sg.lo = (const uint16_t*)L.stats_lo.p;
        sg.ld = stats_ld_;
      } else {
        const BlobLayerInfo& pi = info_.layers[src.layer];
        if (pi.segment_level) {
          sg.hi = (const uint16_t*)L.act[src.layer].act_hi.p;
          sg.lo = (const uint16_t*)L.act[src.layer].act_lo.p;
        } else {
          sg.hi = ActBase(L.act[src.layer].act_hi, pi.n_pad);
          sg.lo = ActBase(L.act[src.layer].act_lo, pi.n_pad);
        }
        sg.ld = pi.n_pad;
      }
      sg.row_shift = li.segment_level ? 0 : src.offset;
      sg.ksteps = SrcKPad(src.dim, src.layer, li.segment_level, info_.precision) / kBK;
      sg.gmax = (mx_pass && !li.segment_level && src.layer >= 0 && !info_.layers[src.layer].segment_level) ? gmax_of(src.layer)
                                                                                                      : nullptr;
      if (fast_mx2_ && !li.segment_level && src.layer >= 0 && !info_.layers[src.layer].segment_level) {
        const int pn = info_.layers[src.layer].n_pad;
        sg.lo4 = (const uint8_t*)L.act[src.layer].act_lo4.p + (size_t)kHalo * (pn / 2);
        sg.lo4s = (const uint8_t*)L.act[src.layer].act_lo4s.p + (size_t)kHalo * Lo4ScalePitch(pn);
      }
      ksteps += sg.ksteps;
    }
    ga.total_ksteps = ksteps;
    ga.w_hi = dl.w_hi;
    ga.w_lo = dl.w_lo;
    ga.ldw = li.k_pad;
    ga.w4 = dl.w4;
    ga.w4_scale = dl.w4_scale;
    ga.ldw4 = dl.ldw4;
    ga.w4b = dl.w4b;
    ga.w4b_scale = dl.w4b_scale;
    ga.ldw4b = dl.ldw4b;
    ga.n_tiles = li.n_pad / kBN;
    ga.relu = li.relu;
    ga.bn = li.bn;
    ga.bias = dl.bias;
    ga.scale = dl.scale;
    ga.offset = dl.offset;
    int epi;
    if (!li.segment_level) {
      ga.m_tiles = plan.rows / kBM;
      if ((int)i == info_.pooled_layer) {
        epi = kEpiStats;
        ga.partial = (float*)L.partial.p;
        ga.ldp = li.n_pad;
        ga.grp_range = plan.d_grp_range;
      } else if (frame_mode_ && (int)i == info_.output_layer && logits16()) {
        epi = kEpiAct;   // the head's logits as an fp16 plane (no halo rows: nothing splices them)
        ga.out_hi = (uint16_t*)L.frame_f32.p;
        ga.out_lo = nullptr;
        ga.ldo = li.n_pad;
      } else if (frame_mode_ && (int)i == info_.output_layer) {
        epi = kEpiF32;
        ga.out_f32 = (float*)L.frame_f32.p;
        ga.ldf = li.n_pad;
        ga.m_valid = plan.rows;
      } else {
        epi = kEpiAct;
        ga.out_hi = ActBase(L.act[i].act_hi, li.n_pad);
        ga.out_lo = ActBase(L.act[i].act_lo, li.n_pad);
        ga.ldo = li.n_pad;
        if (fast_mx2_) {
          ga.out_lo4 = (uint8_t*)L.act[i].act_lo4.p + (size_t)kHalo * (li.n_pad / 2);
          ga.out_lo4s = (uint8_t*)L.act[i].act_lo4s.p + (size_t)kHalo * Lo4ScalePitch(li.n_pad);
        }
      }
    } else {
      ga.m_tiles = plan.b_pad / kBM;
      // few rows, long K (3000 for the embedding layer): K is split over up to 24 slices of at least 4 steps - a rule that does
      // not depend on the batch, so an utterance's sums are formed in the same order whatever it is batched with.  Measured
      // on the 256-chunk embedding layer (94 steps, 2 x 4 tiles), GEMM + reduction: 12 steps x 8 slices 23.3 us, 8 x 12 20.4,
      // 6 x 16 19.4, 4 x 24 18.8, 3 x 32 19.3, 2 x 47 25.2 (with the workgroups spread over all XCDs, kernels.hip; while they
      // all sat on two of them more slices only queued: 25.9 / 31.6 / 36.8 us for 8 / 12 / 24 slices)
      const int per = std::max(4, (ksteps + 23) / 24);
      ga.ksteps_per_slice = per;
      ga.ksplit = (ksteps + per - 1) / per;
      if (ga.ksplit > 1) {
        Ensure(&L.splitk_ws, (size_t)ga.ksplit * plan.b_pad * li.n_pad * 4, false);
        ga.splitk_ws = (float*)L.splitk_ws.p;
      }
      if ((int)i == info_.output_layer) {
        epi = kEpiF32;
        // the embedding goes straight into the caller's buffer when its rows can take the 16-byte stores of all n_pad
        // columns (no padding columns, aligned); otherwise through out_f32 and a strided copy
        direct_out = !li.log_softmax && li.n_pad == info_.output_dim && (out_ld & 3) == 0 && out_ld >= li.n_pad &&
                     ((uintptr_t)out_dev & 15) == 0;
        ga.out_f32 = direct_out ? out_dev : (float*)L.out_f32.p;
        ga.ldf = direct_out ? out_ld : li.n_pad;
        ga.m_valid = plan.B;
      } else {
        epi = kEpiAct;
        ga.out_hi = (uint16_t*)L.act[i].act_hi.p;
        ga.out_lo = (uint16_t*)L.act[i].act_lo.p;
        ga.ldo = li.n_pad;
      }
    }
    // scales of the residual plane in the tile order of this epilogue's operand orientation (TileMxScales)
    if (ga.w4_scale && epi == kEpiStats) ga.w4_scale += (size_t)li.n_pad * (li.k_pad / kBK);
    if (ga.w4b_scale && epi == kEpiStats) ga.w4b_scale += (size_t)li.n_pad * (li.k_pad / kBK);
    arm(std::string("tdnn_gemm<") + (epi == kEpiAct ? "act" : epi == kEpiF32 ? "f32" : "stats") + ">:" + li.name);
    if (dl.first) {
      // reads the caller's fp32 rows through the plan's tables; one launch per row region, the planes epilogue of the
      // region's arithmetic (absolute device rows: no pointer shifts)
      FirstArgs fa;
      memset(&fa, 0, sizeof fa);
      fa.g = ga;
      fa.feats = feats_dev;
      fa.feats_valid_idx = (long)plan.src_off[0] * info_.input_dim;
      fa.grp_src = (const int4*)plan.d_grp_src;
      fa.rows = plan.rows;
      fa.dim = info_.input_dim;
      fa.noff = (int)li.src.size();
      for (int j = 0; j < fa.noff; ++j) fa.off[j] = li.src[j].offset;
      fa.wc_hi = dl.wc_hi;
      fa.wc_lo = dl.wc_lo;
      for (int region = 0; region < 2; ++region) {
        fa.row0 = region == 0 ? 0 : plan.rows_fast;
        fa.nrows = (region == 0 ? plan.rows_fast : plan.rows) - fa.row0;
        if (fa.nrows <= 0) continue;
        int eprec = prec;   // slow region: both planes of the three-pass arithmetic
        fa.g.gmax_out = nullptr;
        fa.g.out_range = nullptr;
        if (region == 0) {
          // (mixed mode: a first layer whose consumers are all lite writes the fp16 plane only)
          eprec = (fast_mx2_ && (!lite_mask_ || lite_emits_[i])) ? (int)kPrecFp16x3E : (int)kPrecFp16x2;
          if (mx_pass) {
            fa.g.gmax_out = gmax_of((int)i);
            fa.g.out_range = plan.d_act_range + (size_t)i * plan.ngrp * 2;
          }
        }
        Check(launch_tdnn_first(fa, eprec, s), "tdnn_first launch");
        if (first_prof) prof_labels_.back() += std::string(" ") + last_gemm_kernel();
      }
    } else if (li.segment_level || plan.rows_fast == 0) {
      if (prec == kPrecFp16 && use_p8_ && !li.segment_level && (epi == kEpiAct || epi == kEpiStats)) {
        // single-pass fp16: the layers tdnn_gemm_kernel_p8 can run, run it - for every launch (see the fast region below)
        GemmArgs g8 = ga;
        g8.p8 = 1;
        g8.p8_whole = p8_whole_;
        if (gemm_p8_applicable(g8, kPrecFp16)) ga = g8;
      }
      Check(launch_tdnn_gemm(ga, prec, epi, s), "tdnn_gemm launch");
      if (first_prof) prof_labels_.back() += std::string(" ") + last_gemm_kernel();
    } else {
      // rows [0, rows_fast): two-pass kernels; the rest: slow_prec_.  Same arguments, row base moved.
      for (int region = 0; region < 2; ++region) {
        const long r0 = region == 0 ? 0 : plan.rows_fast;
        const long r1 = region == 0 ? plan.rows_fast : plan.rows;
        if (r1 <= r0) continue;
        GemmArgs gr = ga;
        gr.m_tiles = (int)((r1 - r0) / kBM);
        for (int j = 0; j < gr.nseg; ++j) {
          gr.seg[j].hi += r0 * gr.seg[j].ld;
          if (gr.seg[j].lo) gr.seg[j].lo += r0 * gr.seg[j].ld;
          if (gr.seg[j].lo4) gr.seg[j].lo4 += r0 * (gr.seg[j].ld / 2);
          if (gr.seg[j].lo4s) gr.seg[j].lo4s += r0 * Lo4ScalePitch(gr.seg[j].ld);
        }
        if (gr.out_hi) gr.out_hi += r0 * gr.ldo;
        if (gr.out_lo) gr.out_lo += r0 * gr.ldo;
        if (gr.out_lo4) gr.out_lo4 += r0 * (gr.ldo / 2);
        if (gr.out_lo4s) gr.out_lo4s += r0 * Lo4ScalePitch(gr.ldo);
        // the fast region records the group maxima of the planes it writes and, where the layer allows, runs the
        // 1.25-pass mode on them (region 0 starts at row 0: the tables need no offset)
        if (region == 0 && mx_pass && epi == kEpiAct) {
          gr.gmax_out = gmax_of((int)i);
          gr.out_range = plan.d_act_range + (size_t)i * plan.ngrp * 2;
        }
        if (region != 0)
          for (int j = 0; j < gr.nseg; ++j) gr.seg[j].gmax = nullptr;
        if (gr.out_f32) gr.out_f32 += r0 * gr.ldf;
        if (gr.partial) gr.partial += (r0 / kRowAlign) * 2 * gr.ldp;
        if (gr.grp_range) gr.grp_range += (r0 / kRowAlign) * 2;
        gr.m_valid = (int)(r1 - r0);
        int rprec = prec;
        if (region == 0) {
          if (fast_mx2_ && lite_mask_ && lite_[i] && mx_pass && gemm_mx_applicable(gr)) {
            // a lite layer of the mixed mode (SetLiteMask): the 1.25-pass product, on the 256 x 256 kernel where no consumer
            // needs the residual plane of its output, else with the planes epilogue that writes it
            if (lite_emits_[i] && epi == kEpiAct) {
              rprec = kPrecFp16MxE;
            } else {
              rprec = kPrecFp16Mx;
              gr.out_lo4 = nullptr;
              gr.out_lo4s = nullptr;
            }
            // (both on the 256 x 256 kernel where the layer's shape allows: the same K walk, so the fp16 plane of a lite layer
            // has the same bits whether or not it also writes its residual plane)
            if (use_p8_ && dl.w4p && (epi == kEpiAct || epi == kEpiStats)) {
              GemmArgs g8 = gr;
              g8.p8 = 1;
              g8.p8_whole = p8_whole_;
              g8.w4 = dl.w4p;
              g8.w4_scale = dl.w4p_scale + (epi == kEpiStats ? (size_t)li.n_pad * (li.k_pad / kBK) : 0);
              if (gemm_p8_applicable(g8, rprec)) gr = g8;
            }
          } else if (fast_mx2_) {
            // every layer emits the 4-bit residual of its fp16 plane; a layer that cannot run the second walk reads the
            // network input only (PackModel checked it) and runs the three-pass arithmetic on the planes of prep_input
            rprec = gemm_mx2_applicable(gr) ? (int)kPrecFp16Mx2 : (int)kPrecFp16x3E;
            if (rprec == kPrecFp16x3E && epi != kEpiAct) throw EngineError("fp16mx2: layer " + li.name + " cannot run the mode");
            if (rprec == kPrecFp16Mx2 && use_p8_ && dl.w4p && dl.w4bp && (epi == kEpiAct || epi == kEpiStats)) {
              // the 1.5-pass launches of the layers tdnn_gemm_kernel_p8 can run: both 4-bit images in its walk orders
              const size_t so = epi == kEpiStats ? (size_t)li.n_pad * (li.k_pad / kBK) : 0;
              GemmArgs g8 = gr;
              g8.p8 = 1;
              g8.p8_whole = p8_whole_;
              g8.w4 = dl.w4p;
              g8.w4_scale = dl.w4p_scale + so;
              g8.w4b = dl.w4bp;
              g8.ldw4b = li.k_pad * 2;
              g8.w4b_scale = dl.w4bp_scale + so;
              if (gemm_p8_applicable(g8, kPrecFp16Mx2)) gr = g8;
            }
          } else {
            rprec = (mx_pass && gemm_mx_applicable(gr)) ? (int)kPrecFp16Mx : (int)kPrecFp16x2;
            if (rprec == kPrecFp16Mx && use_p8_ && dl.w4p && (epi == kEpiAct || epi == kEpiStats)) {
              // the layer's 1.25-pass launches run the 256 x 256 x 64 kernel - all of them, whatever their size: its sums are
              // formed in another order than the 32-column kernels', and a chunk's embedding must not depend on its batch
              GemmArgs g8 = gr;
              g8.p8 = 1;
              g8.p8_whole = p8_whole_;
              g8.w4 = dl.w4p;
              g8.w4_scale = dl.w4p_scale + (epi == kEpiStats ? (size_t)li.n_pad * (li.k_pad / kBK) : 0);
              if (gemm_p8_applicable(g8, kPrecFp16Mx)) gr = g8;
            }
          }
        }
        Check(launch_tdnn_gemm(gr, rprec, epi, s), "tdnn_gemm launch");
        if (first_prof) prof_labels_.back() += std::string(" ") + last_gemm_kernel();
      }
    }
    disarm();

    if ((int)i == info_.pooled_layer) {
      PoolArgs po;
      po.partial = (const float*)L.partial.p;
      po.ldp = li.n_pad;
      po.utt_grp0 = plan.d_utt_grp0;
      po.utt_grp1 = plan.d_utt_grp1;
      po.utt_count = plan.d_utt_count;
      po.B = plan.B;
      po.dim = info_.pool_dim;
      po.var_floor = info_.variance_floor;
      po.out_hi = (uint16_t*)L.stats_hi.p;
      po.out_lo = (uint16_t*)L.stats_lo.p;
      po.ld = stats_ld_;
      arm("pool_finalise");
      Check(launch_pool_finalise(po, prec, s), "pool_finalise launch");
      disarm();
    }
  }
  const BlobLayerInfo& ol = info_.layers[info_.output_layer];
  if (frame_mode_) {
    FrameOutArgs fo;
    fo.src = logits16() ? nullptr : (const float*)L.frame_f32.p;
    fo.src16 = logits16() ? (const uint16_t*)L.frame_f32.p : nullptr;
    fo.ld = ol.n_pad;
    fo.out_row = plan.d_out_row;
    fo.n_out = plan.n_out;
    fo.dim = info_.output_dim;
    fo.log_softmax = ol.log_softmax;
    fo.out = out_dev;
    fo.out_ld = out_ld;
    arm("frame_output");
    Check(launch_frame_output(fo, s), "frame_output launch");
    disarm();
  } else if (ol.log_softmax) {
    // pooled output taken after a LogSoftmaxComponent (e.g. the speaker posteriors of the unedited x-vector net)
    FrameOutArgs fo;
    fo.src = (const float*)L.out_f32.p;
    fo.src16 = nullptr;
    fo.ld = ol.n_pad;
    fo.out_row = nullptr;
    fo.n_out = plan.B;
    fo.dim = info_.output_dim;
    fo.log_softmax = 1;
    fo.out = out_dev;
    fo.out_ld = out_ld;
    arm("frame_output");
    Check(launch_frame_output(fo, s), "frame_output launch");
    disarm();
  } else if (!direct_out) {
    Check(hipMemcpy2DAsync(out_dev, (size_t)out_ld * 4, L.out_f32.p, (size_t)ol.n_pad * 4, (size_t)info_.output_dim * 4,
                           (size_t)plan.B, hipMemcpyDeviceToDevice, s),
          "hipMemcpy2DAsync(out)");
  }
  Check(hipEventRecord(L.done, s), "hipEventRecord(lane)");
  L.busy = true;
  if (prof_on_) prof_runs_.push_back(std::move(prof_run));
}

void Engine::SetFastMode(int mode) {
  if (!can_switch_fast_mode()) throw EngineError("this context cannot switch its arithmetic (it was not packed as fp16mx2 with a pooled output)");
  if (mode != kPrecFp16Mx2 && mode != kPrecFp16Mx && mode != kPrecFp16x3) throw EngineError("SetFastMode: fp16mx2, fp16mx or fp16x3");
  if (mode == fast_mode_ && lite_mask_ == 0) return;   // (the same mode again still clears a mixture)
  Check(hipSetDevice(device_), "hipSetDevice");
  Check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  for (const HostSlot& S : host_slots_)
    if (S.pending) throw EngineError("SetFastMode with a batch in flight");
  fast_mode_ = mode;
  fast_mx2_ = mode == kPrecFp16Mx2;
  fast_mx_ = mode != kPrecFp16x3;
  has_fast_ = mode != kPrecFp16x3;
  fast_min_pooled_ = mode == kPrecFp16Mx2 ? mx2_min_pooled_ : mx_min_pooled_;
  lite_mask_ = 0;
  lite_.assign(layers_.size(), 0);
  lite_emits_.assign(layers_.size(), 0);
  plan_cache_.clear();   // the row regions of a plan depend on the threshold
}

void Engine::SetLiteMask(uint64_t mask) {
  if (!can_switch_fast_mode() || fast_mode_ != kPrecFp16Mx2) throw EngineError("SetLiteMask: the context must be running fp16mx2");
  Check(hipSetDevice(device_), "hipSetDevice");
  Check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  for (const HostSlot& S : host_slots_)
    if (S.pending) throw EngineError("SetLiteMask with a batch in flight");
  const size_t nl = layers_.size();
  lite_.assign(nl, 0);
  lite_emits_.assign(nl, 0);
  uint64_t kept = 0;
  for (size_t i = 0; i < nl && i < (size_t)kMaxLiteLayers; ++i) {
    const BlobLayerInfo& li = info_.layers[i];
    // a layer can be lite when it is a frame-level GEMM over other layers' planes with the residual image of the 1.25-pass
    // arithmetic (the layers that read the network input run the first-layer kernel in its own arithmetic)
    bool ok = ((mask >> i) & 1) && !li.segment_level && li.has_w4 && !layers_[i].first;
    for (const LayerSource& src : li.src) ok = ok && src.layer >= 0 && !info_.layers[src.layer].segment_level;
    if (ok) {
      lite_[i] = 1;
      kept |= 1ull << i;
    }
  }
  // who still has to write the residual plane of its output: a layer with a frame-level consumer that walks it
  for (size_t c = 0; c < nl; ++c) {
    if (info_.layers[c].segment_level || lite_[c]) continue;
    for (const LayerSource& src : info_.layers[c].src)
      if (src.layer >= 0) lite_emits_[src.layer] = 1;
  }
  lite_mask_ = kept;
  fast_min_pooled_ = kept ? std::max(mx2_min_pooled_, mx_min_pooled_) : mx2_min_pooled_;
  plan_cache_.clear();
}

Engine::Calibration Engine::Calibrate(const float* feats, const int32_t* row_offsets, int B, float tol) {
  Calibration c;
  c.chosen = fast_mode();
  if (!can_switch_fast_mode() || B <= 0) return c;
  {
    // fault injection for the tests of the callers' error paths (XVEC_DEBUG=calib_fail=1: the first call of the process fails,
    // 2: every call)
    static std::atomic<int> calls{0};
    const int inject = DebugKnobInt("calib_fail", 0);
    if ((inject == 1 && calls.fetch_add(1) == 0) || inject == 2) throw EngineError("injected failure (XVEC_DEBUG=calib_fail)");
  }
  // (3: the first call of the process sees ONE element of its packed pass move by one ulp - what a device that does not
  // reproduce its own bits looks like to the check below)
  static std::atomic<int> perturb_calls{0};
  const bool perturb = DebugKnobInt("calib_fail", 0) == 3 && perturb_calls.fetch_add(1) == 0;
  // Chunks each fast mode would run fast - pooled frames as FillPlan counts them.  fp16mx2 is validated on every chunk IT
  // runs (from mx2_min_pooled_ frames, where its error is largest), fp16mx on those it runs (from mx_min_pooled_).
  const BlobLayerInfo& pl = info_.layers[info_.pooled_layer];
  const int pool_first = std::max(pl.left, -info_.pool_left);
  std::vector<int32_t> offs(1, row_offsets[0]);
  std::vector<int> pick;
  std::vector<char> runs_mx;
  for (int b = 0; b < B; ++b) {
    const int T = row_offsets[b + 1] - row_offsets[b];
    const int cnt = std::min(T - 1 - pl.right, info_.pool_right) - pool_first + 1;
    if (T >= info_.min_frames && cnt >= std::min(mx_min_pooled_, mx2_min_pooled_)) {
      pick.push_back(b);
      runs_mx.push_back(cnt >= mx_min_pooled_);
    }
  }
  if (pick.empty()) return c;
  // packed copy of the picked chunks
  const int D = info_.input_dim, E = info_.output_dim;
  std::vector<float> f;
  offs.assign(1, 0);
  for (int b : pick) {
    f.insert(f.end(), feats + (size_t)row_offsets[b] * D, feats + (size_t)row_offsets[b + 1] * D);
    offs.push_back(offs.back() + (row_offsets[b + 1] - row_offsets[b]));
  }
  const int n = (int)pick.size();
  const int before = fast_mode_;
  const uint64_t lite_before = lite_mask_;
  std::vector<float> ref((size_t)n * E), mx((size_t)n * E), mx2((size_t)n * E);
  try {
    SetFastMode(kPrecFp16x3);
    ForwardHost(f.data(), offs.data(), n, ref.data());
    SetFastMode(kPrecFp16Mx);
    ForwardHost(f.data(), offs.data(), n, mx.data());
    SetFastMode(kPrecFp16Mx2);
    ForwardHost(f.data(), offs.data(), n, mx2.data());
    if (perturb) mx2[0] = std::nextafterf(mx2[0], INFINITY);
    // The measurement is only worth something if the device computes the same bits twice.  The three passes above are the FIRST
    // forward passes of a job's context (fresh planes, first launches, and - run.pl JOB=1:nj - three other processes doing
    // the same on the same GPU); the reference and the packed arithmetic are run once more and compared byte for byte.  A
    // difference is a device or library fault and ends the job with a message instead of moving a threshold decision
    // (r05: one of four concurrent jobs on the driver's box came out of its calibration with another arithmetic than the
    // solo run, and nothing in its log said why - DESIGN.md section 0).
    std::vector<float> again((size_t)n * E);
    for (int pass = 0; pass < 2; ++pass) {
      const std::vector<float>& first = pass == 0 ? mx2 : ref;
      if (pass == 1) SetFastMode(kPrecFp16x3);
      ForwardHost(f.data(), offs.data(), n, again.data());
      if (memcmp(again.data(), first.data(), again.size() * sizeof(float)) != 0) {
        int bad = 0;
        double worst = 0.0;
        for (int i = 0; i < n; ++i) {
          bool differs = false;
          for (int k = 0; k < E; ++k) {
            const float a = again[(size_t)i * E + k], b = first[(size_t)i * E + k];
            if (memcmp(&a, &b, 4) != 0) {
              differs = true;
              const double r = std::fabs((double)a - b) / std::max(1e-30, (double)std::fabs(b));
              if (!(r <= worst)) worst = r;
            }
          }
          bad += differs ? 1 : 0;
        }
        std::ostringstream m;
        m << "calibration: two runs of the " << PrecisionName(pass == 0 ? (int)kPrecFp16Mx2 : (int)kPrecFp16x3) << " arithmetic on the same "
          << n << " chunks differ in " << bad << " of them (largest relative difference of an element " << worst
          << "): the device does not reproduce its own results";
        throw EngineError(m.str());
      }
    }
  } catch (...) {
    SetFastMode(before);
    if (before == kPrecFp16Mx2 && lite_before) SetLiteMask(lite_before);
    throw;
  }
  // Per-chunk errors of one arithmetic over a set of sample chunks.  half: -1 = every chunk, 0 / 1 = the chunks at even / odd
  // positions of the sample (selection / held-out half of the mixture).
  // A configuration is ACCEPTED on a set when (a) its worst chunk there is within the tolerance and (b) the tail its error
  // distribution projects - mean + kTailSigmas (6) standard deviations over the set - is within tol x kTailOverTol (1.10: the
  // tools' 7.5e-5 -> 8.25e-5).  (b) is there because a job is not 64 chunks.  Measured over 32 768 distinct chunks per model
  // (tools/tail_error.py, profiles/r05_tail_error.md): the worst chunk lies 4.6-5.9 standard deviations above the mean; the
  // projection from the 64-chunk sample landed 1-7 % above that worst chunk for fp16mx / plain fp16mx2 and 6-10 %
  // BELOW it for mixtures (32 confirming chunks, a selected configuration) - and a mixture of the c-vector network that sat at
  // 6.9e-5 on its sample (mean 5.7e-5, projection 9.0e-5) reached 1.0008e-4 against the fp64 oracle on one chunk of 32 768.
  // Hence a limit of 8.25e-5 for the projection rather than the bar itself: the 10 % it has been seen to fall short, and a
  // job thirty times that size.  (At 1.15 x one of twenty-one models - fp16mx accepted with a projection of 8.62e-5 - reached
  // 9.4e-5 in 8192 chunks; at 1.10 x the worst chunk of all of them is 8.5e-5.)  A candidate whose errors are spread too wide for its mean is turned down even when none of
  // the sampled chunks is above the tolerance.  Sets of fewer than eight chunks: (a) only.
  struct ErrStat {
    float worst = 0.f, tail = 0.f;
    int n = 0;
  };
  auto stat = [&](const std::vector<float>& got, bool only_mx, int half) {
    ErrStat st;
    double sum = 0.0, sum2 = 0.0;
    for (int i = 0; i < n; ++i) {
      if (only_mx && !runs_mx[i]) continue;
      if (half >= 0 && (i & 1) != half) continue;
      float d = 0.f, m = 0.f;
      for (int k = 0; k < E; ++k) {
        d = std::max(d, std::fabs(got[(size_t)i * E + k] - ref[(size_t)i * E + k]));
        m = std::max(m, std::fabs(ref[(size_t)i * E + k]));
      }
      const float r = m > 0.f ? d / m : (d > 0.f ? INFINITY : 0.f);
      st.worst = std::isnan(r) ? INFINITY : std::max(st.worst, r);
      sum += r;
      sum2 += (double)r * r;
      ++st.n;
    }
    st.tail = st.worst;
    if (st.n >= 8 && std::isfinite(st.worst)) {
      const double mean = sum / st.n, var = std::max(0.0, (sum2 - st.n * mean * mean) / (st.n - 1));
      st.tail = std::max(st.worst, (float)(mean + kTailSigmas * std::sqrt(var)));
    }
    return st;
  };
  // (XVEC_DEBUG=tail_over_tol=<factor>: diagnostic override, for studies like tools/tail_error.py)
  float tail_over_tol = kTailOverTol;
  {
    const std::string e = DebugKnob("tail_over_tol");
    if (!e.empty() && atof(e.c_str()) >= 1.0) tail_over_tol = (float)atof(e.c_str());
  }
  auto accepted = [&](const ErrStat& st) { return st.worst <= tol && st.tail <= tol * tail_over_tol; };
  int n_mx = 0, n_mx_hold = 0;
  for (int i = 0; i < n; ++i) {
    n_mx += runs_mx[i] ? 1 : 0;
    n_mx_hold += (runs_mx[i] && (i & 1)) ? 1 : 0;
  }
  c.checked = n;
  c.checked_mx = n_mx;
  const ErrStat st_mx = stat(mx, true, -1), st_mx2 = stat(mx2, false, -1);
  c.err_mx = st_mx.worst;
  c.err_mx2 = st_mx2.worst;
  c.tail_mx = st_mx.tail;
  // The lighter mode governs a whole job: it is taken only on the evidence of at least kCalibMinChunks chunks it would run
  // (the margin between the tolerance and the bar is argued for a sample of that order; one or two
  // qualifying chunks are not a measurement).  Fewer: the packed mode stays.  This is ONE hypothesis fixed in advance,
  // tested once on the whole sample - nothing is selected, so nothing has to be held out.
  const bool mx_ok = n_mx >= kCalibMinChunks && accepted(st_mx);
  // (the packed 1.5-pass mode is left only for the three-pass one, at half the speed: it goes when a sampled chunk or the projected
  // tail is too close to the BAR - on the heavy-tailed c-vector model with chunks of 175-600 frames it projects 8.9e-5 and its worst
  // of 32 768 chunks is 8.8e-5)
  // (the projection has been seen up to 16 % below the worst of 32 768 chunks - profiles/r05_tail_error.md, 33 configurations - so
  // the limit for staying in the packed mode is tol x kTailOverTolPacked = 9e-5, not the bar)
  c.chosen = mx_ok ? (int)kPrecFp16Mx
                   : ((c.err_mx2 <= 1e-4f && st_mx2.tail <= tol * kTailOverTolPacked) ? (int)kPrecFp16Mx2 : (int)kPrecFp16x3);
  c.tail = mx_ok ? st_mx.tail : (c.chosen == kPrecFp16Mx2 ? st_mx2.tail : 0.f);
  SetFastMode(c.chosen);
  // (the mixture runs fast what fp16mx runs fast - chunks of >= 300 pooled frames - and sends the shorter ones, which plain
  // fp16mx2 runs fast from 160, to the three-pass arithmetic: on a job with many of those it would lose more than it saves,
  // so it is considered only where they are at most an eighth of the sample)
  const int n_mx_fit = n_mx - n_mx_hold;
  if (c.chosen == kPrecFp16Mx2 && n_mx_fit >= kCalibMinChunks / 2 && n_mx_hold >= kCalibMinChunks / 2 && accepted(st_mx2) &&
      (n - n_mx) * 8 <= n) {
    // Between the two: the 1.5-pass context with some of its layers in 1.25 passes - as much of the second walk taken off as the
    // tolerance allows.  What the second walk corrects (the activations' fp16 rounding) matters less the further a layer is
    // from the pooled statistics: on the c-vector network the 650-wide phonetic branch hardly needs it, the x-vector branch
    // does.  This IS a selection - up to 2 x (candidate layers) configurations are tried and the cheapest that passes is
    // kept, so the error of the winner on the chunks that chose it is biased low.  Hence two halves:
    //   selection half (even positions): every candidate layer measured alone, ranked by second-walk work saved
    //     (k_pad x n_pad) per unit of squared error added (the excesses add up roughly in squares), then greedily in that
    //     ranking - a layer joins if the mixture with it is still within the tolerance on THIS half;
    //   held-out half (odd positions): the adopted mixture must be within the tolerance here as well; while it is not, the
    //     layer added last is taken out again (fp16mx2 itself, the empty mixture, passed on the whole sample above).
    // Every forward pass runs the whole sample (the halves share the batch), only the comparison is restricted.
    std::vector<int> order;
    for (size_t i = 0; i < layers_.size() && i < (size_t)kMaxLiteLayers; ++i) {
      const BlobLayerInfo& li = info_.layers[i];
      bool ok = !li.segment_level && li.has_w4 && !layers_[i].first;
      for (const LayerSource& src : li.src) ok = ok && src.layer >= 0 && !info_.layers[src.layer].segment_level;
      if (ok) order.push_back((int)i);
    }
    std::vector<float> got((size_t)n * E);
    float err_all = 0.f, err_hold = 0.f, tail_hold = 0.f;
    int dropped = 0;
    try {
      std::vector<double> gain(layers_.size(), 0.0);
      const double fit_mx2 = stat(mx2, true, 0).worst;
      const double base2 = fit_mx2 * fit_mx2;
      for (int i : order) {
        SetLiteMask(1ull << i);
        if (!lite_mask_) continue;
        ForwardHost(f.data(), offs.data(), n, got.data());
        const ErrStat sa = stat(got, true, 0);
        const double e = sa.worst;
        if (!accepted(sa)) continue;   // not even alone
        const double v = std::max(e * e - base2, 1e-4 * base2 + 1e-30);
        gain[i] = (double)info_.layers[i].k_pad * info_.layers[i].n_pad / v;
      }
      order.erase(std::remove_if(order.begin(), order.end(), [&](int i) { return gain[i] <= 0.0; }), order.end());
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gain[a] > gain[b]; });
      // in that order, a layer joins the mixture if the mixture with it - run on the sample - is still within the tolerance
      // on the selection half (one pass per candidate; a layer that does not fit is skipped, the cheaper ones behind it
      // still get their turn)
      uint64_t mask = 0;
      std::vector<int> added;
      for (int i : order) {
        SetLiteMask(mask | (1ull << i));
        if (lite_mask_ == mask) continue;
        ForwardHost(f.data(), offs.data(), n, got.data());
        if (accepted(stat(got, true, 0))) {
          mask = lite_mask_;
          added.push_back(i);
        }
      }
      // confirmation on the half that did not choose
      while (mask) {
        SetLiteMask(mask);
        ForwardHost(f.data(), offs.data(), n, got.data());
        const ErrStat sh = stat(got, true, 1);
        err_hold = sh.worst;
        tail_hold = sh.tail;
        err_all = std::max(err_hold, stat(got, true, 0).worst);
        if (accepted(sh)) break;
        mask &= ~(1ull << added.back());
        added.pop_back();
        ++dropped;
        err_hold = err_all = tail_hold = 0.f;
      }
      SetLiteMask(mask);
    } catch (...) {
      SetFastMode(before);
      if (before == kPrecFp16Mx2 && lite_before) SetLiteMask(lite_before);
      throw;
    }
    c.lite_mask = lite_mask_;
    c.err_lite = lite_mask_ ? err_all : 0.f;
    c.err_holdout = lite_mask_ ? err_hold : 0.f;
    c.checked_holdout = n_mx_hold;
    c.lite_dropped = dropped;
    if (lite_mask_) c.tail = tail_hold;
  }
  return c;
}

std::string Engine::ProfileReport() {
  Check(hipSetDevice(device_), "hipSetDevice");
  Check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  std::vector<double> tot(prof_labels_.size(), 0.0);
  for (auto& run : prof_runs_) {
    for (size_t i = 0; 2 * i + 1 < run.size() && i < tot.size(); ++i) {   // (start, stop) pair of launch i
      float ms = 0.f;
      Check(hipEventElapsedTime(&ms, run[2 * i], run[2 * i + 1]), "hipEventElapsedTime");
      tot[i] += ms;
    }
    for (hipEvent_t e : run) (void)hipEventDestroy(e);
  }
  std::ostringstream o;
  o.precision(9);
  for (size_t i = 0; i < tot.size(); ++i) o << prof_labels_[i] << "\t" << prof_runs_.size() << "\t" << tot[i] << "\n";
  prof_runs_.clear();
  return o.str();
}

void Engine::FrontEndHost(const float* raw, const int32_t* raw_off, int n_utts, const int32_t* sel_row,
                          const int32_t* sel_utt, int n_out, int cmn_window, bool center, int min_window, float* out) {
  Check(hipSetDevice(device_), "hipSetDevice");
  if (n_out <= 0) return;
  const int D = info_.input_dim;
  const long raw_rows = raw_off[n_utts];
  Check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  Ensure(&fe_raw_, (size_t)raw_rows * D * 4, false);
  Ensure(&fe_prefix_, (size_t)(raw_rows + n_utts) * D * 8, false);
  Ensure(&fe_out_, (size_t)n_out * D * 4, false);
  const size_t o_off = 0, o_row = Align256((size_t)(n_utts + 1) * 4), o_utt = Align256(o_row + (size_t)n_out * 4);
  const size_t tab = Align256(o_utt + (size_t)n_out * 4);
  Ensure(&fe_tab_, tab, false);
  std::vector<uint8_t> host(tab, 0);
  memcpy(host.data() + o_off, raw_off, (size_t)(n_utts + 1) * 4);
  memcpy(host.data() + o_row, sel_row, (size_t)n_out * 4);
  memcpy(host.data() + o_utt, sel_utt, (size_t)n_out * 4);
  Check(hipMemcpyAsync(fe_tab_.p, host.data(), tab, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(front-end tables)");
  Check(hipMemcpyAsync(fe_raw_.p, raw, (size_t)raw_rows * D * 4, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(raw feats)");
  FrontEndArgs fa;
  fa.raw = (const float*)fe_raw_.p;
  fa.raw_off = (const int32_t*)((const uint8_t*)fe_tab_.p + o_off);
  fa.prefix = (double*)fe_prefix_.p;
  fa.n_utts = n_utts;
  fa.dim = D;
  fa.sel_row = (const int32_t*)((const uint8_t*)fe_tab_.p + o_row);
  fa.sel_utt = (const int32_t*)((const uint8_t*)fe_tab_.p + o_utt);
  fa.n_out = n_out;
  fa.cmn_window = cmn_window;
  fa.center = center ? 1 : 0;
  fa.min_window = min_window;
  fa.out = (float*)fe_out_.p;
  Check(launch_frontend(fa, stream_), "front-end launch");
  Check(hipMemcpyAsync(out, fe_out_.p, (size_t)n_out * D * 4, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(front-end out)");
  Check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

void Engine::EnsurePinned(void** p, size_t* have, size_t bytes) {
  if (*have >= bytes && *p) return;
  if (*p) Check(hipHostFree(*p), "hipHostFree");
  *p = nullptr;
  *have = 0;
  const size_t want = bytes + bytes / 4 + 4096;   // grow with some slack: batches differ by a few rows
  Check(hipHostMalloc(p, want, hipHostMallocDefault), "hipHostMalloc");
  *have = want;
}

float* Engine::HostFeats(int slot, size_t rows) {
  if (slot < 0 || slot >= kNumHostSlots) throw EngineError("bad host slot");
  HostSlot& S = host_slots_[slot];
  if (S.pending) throw EngineError("host slot reused before WaitHost");
  Check(hipSetDevice(device_), "hipSetDevice");
  EnsurePinned(&S.h_feats, &S.h_feats_bytes, std::max(rows, (size_t)1) * info_.input_dim * 4);
  return (float*)S.h_feats;
}

void Engine::SubmitHost(int slot, long seq, const int32_t* row_offsets, int B, const FrontEndJob* fe) {
  if (slot < 0 || slot >= kNumHostSlots) throw EngineError("bad host slot");
  HostSlot& S = host_slots_[slot];
  if (S.pending) throw EngineError("host slot reused before WaitHost");
  if (row_offsets[0] != 0) throw EngineError("SubmitHost: row_offsets[0] must be 0");
  Check(hipSetDevice(device_), "hipSetDevice");
  const size_t lane = (size_t)seq % lanes_.size();
  S.lane = (int)lane;
  hipStream_t s = lanes_[lane].stream;
  if (!S.done) Check(hipEventCreateWithFlags(&S.done, hipEventDisableTiming), "hipEventCreate(slot)");
  if (!S.h2d_done) Check(hipEventCreateWithFlags(&S.h2d_done, hipEventDisableTiming), "hipEventCreate(slot)");
  if (!copy_stream_) Check(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking), "hipStreamCreate(copy)");
  // The uploads go on their own stream: the lane may still be busy with an older batch, and queued behind it the
  // 12 MB copy would only start when that batch ends and the lane would then idle for the length of the copy.
  hipStream_t cs = copy_stream_;
  S.plan.reset(new Plan());
  std::vector<uint8_t> tables;
  FillPlan(row_offsets, B, S.plan.get(), &tables);
  const size_t fbytes = (size_t)row_offsets[B] * info_.input_dim * 4;
  const size_t obytes = (size_t)(frame_mode_ ? S.plan->n_out : B) * info_.output_dim * 4;
  const bool fe_cm = fe && fe->cm_off != nullptr;
  // what the pinned buffer holds: the batch's chunks, raw float rows for the front-end, or compressed objects for it
  const size_t hbytes = fe_cm ? fe->cm_bytes : fe ? (size_t)fe->raw_off[fe->n_utts] * info_.input_dim * 4 : fbytes;
  if (hbytes > S.h_feats_bytes) throw EngineError("SubmitHost: the batch is larger than the buffer HostFeats returned");
  if (fe && fe->n_out != row_offsets[B]) throw EngineError("SubmitHost: front-end output rows do not match row_offsets");
  // (re)allocation frees device memory, which waits for the device: buffers only grow
  auto grow = [&](Buf* b, size_t need) {
    if (b->bytes < need || !b->p) Ensure(b, need + need / 4 + 256, false);
  };
  grow(&S.d_feats, fbytes);
  grow(&S.d_out, obytes);
  grow(&S.d_tables, tables.size());
  EnsurePinned(&S.h_tables, &S.h_tables_bytes, tables.size());
  EnsurePinned(&S.h_out, &S.h_out_bytes, std::max(obytes, (size_t)4));
  memcpy(S.h_tables, tables.data(), tables.size());
  S.plan->d_tables = S.d_tables.p;
  S.plan->borrowed_tables = true;
  BindPlan(S.plan.get(), S.d_tables.p);
  Check(hipMemcpyAsync(S.d_tables.p, S.h_tables, tables.size(), hipMemcpyHostToDevice, cs), "hipMemcpyAsync(plan tables)");
  if (!fe) {
    Check(hipMemcpyAsync(S.d_feats.p, S.h_feats, fbytes, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(feats)");
    Check(hipEventRecord(S.h2d_done, cs), "hipEventRecord(upload)");
    Check(hipStreamWaitEvent(s, S.h2d_done, 0), "hipStreamWaitEvent(upload)");
  } else {
    const int D = info_.input_dim;
    const long raw_rows = fe->raw_off[fe->n_utts];
    const size_t o_off = 0, o_row = Align256((size_t)(fe->n_utts + 1) * 4), o_utt = Align256(o_row + (size_t)fe->n_out * 4);
    const size_t o_cm = Align256(o_utt + (size_t)fe->n_out * 4);
    const size_t tab = Align256(o_cm + (fe_cm ? (size_t)fe->n_utts * 8 : 0));
    grow(&S.d_raw, (size_t)raw_rows * D * 4);
    grow(&S.d_prefix, (size_t)(raw_rows + fe->n_utts) * D * 8);
    grow(&S.d_fetab, tab);
    EnsurePinned(&S.h_fetab, &S.h_fetab_bytes, tab);
    uint8_t* ht = (uint8_t*)S.h_fetab;
    memcpy(ht + o_off, fe->raw_off, (size_t)(fe->n_utts + 1) * 4);
    memcpy(ht + o_row, fe->sel_row, (size_t)fe->n_out * 4);
    memcpy(ht + o_utt, fe->sel_utt, (size_t)fe->n_out * 4);
    if (fe_cm) memcpy(ht + o_cm, fe->cm_off, (size_t)fe->n_utts * 8);
    Check(hipMemcpyAsync(S.d_fetab.p, S.h_fetab, tab, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(front-end tables)");
    if (fe_cm) {
      grow(&S.d_cm, hbytes);
      Check(hipMemcpyAsync(S.d_cm.p, S.h_feats, hbytes, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(compressed feats)");
    } else {
      Check(hipMemcpyAsync(S.d_raw.p, S.h_feats, hbytes, hipMemcpyHostToDevice, cs), "hipMemcpyAsync(raw feats)");
    }
    Check(hipEventRecord(S.h2d_done, cs), "hipEventRecord(upload)");
    Check(hipStreamWaitEvent(s, S.h2d_done, 0), "hipStreamWaitEvent(upload)");
    if (fe_cm) {   // one byte per element came up; the float rows are made here
      CmExpandArgs ca;
      ca.cm = (const uint8_t*)S.d_cm.p;
      ca.cm_off = (const int64_t*)((const uint8_t*)S.d_fetab.p + o_cm);
      ca.raw_off = (const int32_t*)((const uint8_t*)S.d_fetab.p + o_off);
      ca.n_utts = fe->n_utts;
      ca.dim = D;
      ca.max_rows = fe->max_rows;
      ca.out = (float*)S.d_raw.p;
      Check(launch_cm_expand(ca, s), "cm_expand launch");
    }
    FrontEndArgs fa;
    fa.raw = (const float*)S.d_raw.p;
    fa.raw_off = (const int32_t*)((const uint8_t*)S.d_fetab.p + o_off);
    fa.prefix = (double*)S.d_prefix.p;
    fa.n_utts = fe->n_utts;
    fa.dim = D;
    fa.sel_row = (const int32_t*)((const uint8_t*)S.d_fetab.p + o_row);
    fa.sel_utt = (const int32_t*)((const uint8_t*)S.d_fetab.p + o_utt);
    fa.n_out = fe->n_out;
    fa.cmn_window = fe->cmn_window;
    fa.center = fe->center ? 1 : 0;
    fa.min_window = fe->min_window;
    fa.out = (float*)S.d_feats.p;
    Check(launch_frontend(fa, s), "front-end launch");
  }
  ForwardOnLane(lane, *S.plan, (const float*)S.d_feats.p, (float*)S.d_out.p, info_.output_dim, s);
  Check(hipMemcpyAsync(S.h_out, S.d_out.p, obytes, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(out)");
  Check(hipEventRecord(S.done, s), "hipEventRecord(slot)");
  S.pending = true;
}

void Engine::CheckKernelFaults(int lane) const {
  // a stream-K workgroup that gave up waiting for another workgroup's partial tile (bounded spin, kernels.hip) left
  // a word behind on its launch stream; the results of that launch are not to be trusted.  The word belongs to the
  // stream, i.e. to this engine, and reading it clears it: the caller may retry (XVEC_DEBUG=gemm_variant=2) or exit cleanly.
  // The words of all streams are collected into fault_mask_, but a caller that waited for ONE lane's batch is only told about
  // that lane: a fault of the batch still in flight on the other lane stays recorded until its own WaitHost.
  (void)hipSetDevice(device_);
  if (sk_take_error(stream_)) fault_mask_ |= 1u << 31;
  static_assert(kMaxLanes <= 30, "fault_mask_: bits 0..29 lanes, 30 callers' streams, 31 the engine's stream");
  for (size_t i = 0; i < lanes_.size(); ++i)
    if (sk_take_error(lanes_[i].stream)) fault_mask_ |= 1u << i;
  // launches on a caller's stream (xv_forward_batch_device) leave their word with that stream: read when the caller asks
  // about everything (xv_ctx_synchronize), never on behalf of one lane's batch
  if (lane < 0)
    for (hipStream_t s : ext_streams_)
      if (sk_take_error(s)) fault_mask_ |= 1u << 30;
  unsigned err;
  if (lane < 0) {
    err = fault_mask_;
    fault_mask_ = 0;
  } else {
    const unsigned bits = (1u << lane) | (1u << 31);   // lane < kMaxLanes
    err = fault_mask_ & bits;
    fault_mask_ &= ~bits;
  }
  if (err)
    throw EngineError("a stream-K GEMM launch timed out waiting for a partial tile of another workgroup (results invalid); "
                      "XVEC_DEBUG=gemm_variant=2 selects the per-tile kernels");
}

const float* Engine::WaitHost(int slot) {
  if (slot < 0 || slot >= kNumHostSlots) throw EngineError("bad host slot");
  HostSlot& S = host_slots_[slot];
  if (!S.pending) throw EngineError("WaitHost: nothing submitted on this slot");
  // The wait SLEEPS between polls.  hipEventSynchronize spins - a whole core per process for as long as the GPU is the slower side
  // (3 us of a 9 us host budget per utterance; an event created with hipEventBlockingSync measured the same) - and with three
  // batches queued per engine a wake-up some tens of microseconds late is never on the GPU's critical path: eight ranks of a node
  // share its cores.  XVEC_DEBUG=spin_wait=1: the runtime's wait.
  static const bool spin = [] {
    return DebugKnobInt("spin_wait", 0) != 0;
  }();
  if (spin) {
    Check(hipEventSynchronize(S.done), "hipEventSynchronize(slot)");
  } else {
    for (;;) {
      const hipError_t q = hipEventQuery(S.done);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) Check(q, "hipEventQuery(slot)");
      struct timespec ts = {0, 40000};
      nanosleep(&ts, nullptr);
    }
  }
  S.pending = false;   // before the fault check: the slot is free again whatever the launch reported
  CheckKernelFaults(S.lane);
  return (const float*)S.h_out;
}

void Engine::ForwardHost(const float* feats, const int32_t* row_offsets, int B, float* out) {
  Check(hipSetDevice(device_), "hipSetDevice");
  std::shared_ptr<Plan> plan = MakePlan(row_offsets, B);
  const size_t fbytes = (size_t)(row_offsets[B] - row_offsets[0]) * info_.input_dim * 4;
  const size_t obytes = (size_t)(frame_mode_ ? plan->n_out : B) * info_.output_dim * 4;
  if (feats_stage_.bytes < fbytes || out_stage_.bytes < obytes) Check(hipStreamSynchronize(stream_), "sync");
  Ensure(&feats_stage_, std::max(fbytes, (size_t)4), false);
  Ensure(&out_stage_, obytes, false);
  // the device copy starts at the first row the caller referenced
  Check(hipMemcpyAsync(feats_stage_.p, feats + (size_t)row_offsets[0] * info_.input_dim, fbytes, hipMemcpyHostToDevice,
                       stream_),
        "hipMemcpyAsync(feats)");
  // plan offsets are absolute; shift the device base so that offset row_offsets[0] lands on byte 0
  const float* dev_feats = (const float*)feats_stage_.p - (size_t)row_offsets[0] * info_.input_dim;
  Forward(*plan, dev_feats, (float*)out_stage_.p, info_.output_dim, stream_);
  Check(hipMemcpyAsync(out, out_stage_.p, obytes, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(out)");
  Check(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  CheckKernelFaults();
}

}  // namespace xv
