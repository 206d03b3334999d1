/* TEST INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).
 *
 * Plain-C fp32 restatement of the forward pass that Kaldi's nnet3-xvector-compute performs for the graphs
 * the reference defines (egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:94-114 and the c-vector variants),
 * written as straightforward loops over a flat layer list (exported by oracle/export_program.py):
 *
 *   z_l[t] = W_l . concat_j y_{src_j}[t + off_j] + b_l      only for frames whose whole receptive field exists
 *   y_l[t] = s_l * max(z_l[t], 0) + c_l                      (.affine -> .relu -> .batchnorm, test mode)
 *   mu = mean_t y_L[t];  sigma = sqrt(max(E[y^2] - mu^2, floor));  e = W_6 . [mu; sigma] + b_6
 *   (SURVEY.md App. B.7).  Kaldi itself is not vendored by the reference and cannot be compiled here.
 *
 * It is a third, independent formulation next to the numpy graph evaluator (oracle/xvector_oracle.py) and the
 * torch-conv1d one (tests/golden/make_numeric_goldens.py); tests compare them.  Never linked into the product.
 *
 * usage: xvec_oracle_c <program.bin> <feats.f32> <T> <out.f32>       one chunk, writes the embedding
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_SRC 8

typedef struct {
  int32_t nsrc, src[MAX_SRC], off[MAX_SRC], dim[MAX_SRC];
  int32_t in_dim, out_dim, relu, bn, segment;
  float *wt; /* transposed: [in_dim][out_dim] */
  float *bias, *scale, *offset;
  int lo, right; /* frames lost on the left / right */
  float* y;      /* [(T - lo - right)][out_dim] or [1][out_dim] */
} layer_t;

static void die(const char* m) {
  fprintf(stderr, "xvec_oracle_c: %s\n", m);
  exit(2);
}

static void rd(void* p, size_t n, FILE* f) {
  if (fread(p, 1, n, f) != n) die("short read");
}

int main(int argc, char** argv) {
  if (argc != 5) die("usage: xvec_oracle_c <program.bin> <feats.f32> <T> <out.f32>");
  FILE* f = fopen(argv[1], "rb");
  if (!f) die("cannot open program");
  char magic[8];
  rd(magic, 8, f);
  if (memcmp(magic, "XVORACLE", 8)) die("bad program magic");
  int32_t input_dim, n_layers, pooled, output;
  float var_floor;
  rd(&input_dim, 4, f);
  rd(&n_layers, 4, f);
  rd(&pooled, 4, f);
  rd(&output, 4, f);
  rd(&var_floor, 4, f);
  layer_t* L = (layer_t*)calloc((size_t)n_layers, sizeof(layer_t));
  for (int l = 0; l < n_layers; ++l) {
    rd(&L[l].nsrc, 4, f);
    for (int j = 0; j < L[l].nsrc; ++j) {
      rd(&L[l].src[j], 4, f);
      rd(&L[l].off[j], 4, f);
      rd(&L[l].dim[j], 4, f);
    }
    rd(&L[l].in_dim, 4, f);
    rd(&L[l].out_dim, 4, f);
    rd(&L[l].relu, 4, f);
    rd(&L[l].bn, 4, f);
    rd(&L[l].segment, 4, f);
    const size_t K = (size_t)L[l].in_dim, N = (size_t)L[l].out_dim;
    float* w = (float*)malloc(K * N * 4);
    rd(w, K * N * 4, f); /* [N][K] row-major, Kaldi <LinearParams> orientation */
    L[l].wt = (float*)malloc(K * N * 4);
    for (size_t n = 0; n < N; ++n)
      for (size_t k = 0; k < K; ++k) L[l].wt[k * N + n] = w[n * K + k];
    free(w);
    L[l].bias = (float*)malloc(N * 4);
    L[l].scale = (float*)malloc(N * 4);
    L[l].offset = (float*)malloc(N * 4);
    rd(L[l].bias, N * 4, f);
    rd(L[l].scale, N * 4, f);
    rd(L[l].offset, N * 4, f);
  }
  fclose(f);

  const int T = atoi(argv[3]);
  float* x = (float*)malloc((size_t)T * input_dim * 4);
  f = fopen(argv[2], "rb");
  if (!f) die("cannot open features");
  rd(x, (size_t)T * input_dim * 4, f);
  fclose(f);

  float* stats = NULL;
  for (int l = 0; l < n_layers; ++l) {
    layer_t* a = &L[l];
    const int N = a->out_dim;
    if (!a->segment) {
      a->lo = 0;
      a->right = 0;
      for (int j = 0; j < a->nsrc; ++j) {
        const int sl = a->src[j] < 0 ? 0 : L[a->src[j]].lo, sr = a->src[j] < 0 ? 0 : L[a->src[j]].right;
        if (sl - a->off[j] > a->lo) a->lo = sl - a->off[j];
        if (sr + a->off[j] > a->right) a->right = sr + a->off[j];
      }
      const int frames = T - a->lo - a->right;
      if (frames < 1) die("chunk shorter than the network context (nnet3 never pads)");
      a->y = (float*)malloc((size_t)frames * N * 4);
      for (int i = 0; i < frames; ++i) {
        const int t = a->lo + i;
        float* z = a->y + (size_t)i * N;
        memcpy(z, a->bias, (size_t)N * 4);
        int k0 = 0;
        for (int j = 0; j < a->nsrc; ++j) {
          const float* s;
          if (a->src[j] < 0) s = x + (size_t)(t + a->off[j]) * input_dim;
          else s = L[a->src[j]].y + (size_t)(t + a->off[j] - L[a->src[j]].lo) * L[a->src[j]].out_dim;
          for (int d = 0; d < a->dim[j]; ++d) {
            const float v = s[d];
            const float* w = a->wt + (size_t)(k0 + d) * N;
            for (int n = 0; n < N; ++n) z[n] += v * w[n];
          }
          k0 += a->dim[j];
        }
        for (int n = 0; n < N; ++n) {
          float v = z[n];
          if (a->relu) v = v > 0.f ? v : 0.f;
          if (a->bn) v = v * a->scale[n] + a->offset[n];
          z[n] = v;
        }
      }
      if (l == pooled) {
        stats = (float*)malloc((size_t)2 * N * 4);
        for (int n = 0; n < N; ++n) {
          float s1 = 0.f, s2 = 0.f;
          for (int i = 0; i < frames; ++i) {
            const float v = a->y[(size_t)i * N + n];
            s1 += v;
            s2 += v * v;
          }
          const float mu = s1 / (float)frames;
          volatile float m2 = mu * mu; /* separately rounded, like Kaldi's AddVecVec */
          float var = s2 / (float)frames - m2;
          if (var < var_floor) var = var_floor;
          stats[n] = mu;
          stats[N + n] = sqrtf(var);
        }
      }
    } else {
      a->y = (float*)malloc((size_t)N * 4);
      memcpy(a->y, a->bias, (size_t)N * 4);
      int k0 = 0;
      for (int j = 0; j < a->nsrc; ++j) {
        const float* s = a->src[j] == -2 ? stats : L[a->src[j]].y;
        if (!s) die("pooled statistics used before the pooled layer");
        for (int d = 0; d < a->dim[j]; ++d) {
          const float v = s[d];
          const float* w = a->wt + (size_t)(k0 + d) * N;
          for (int n = 0; n < N; ++n) a->y[n] += v * w[n];
        }
        k0 += a->dim[j];
      }
      for (int n = 0; n < N; ++n) {
        float v = a->y[n];
        if (a->relu) v = v > 0.f ? v : 0.f;
        if (a->bn) v = v * a->scale[n] + a->offset[n];
        a->y[n] = v;
      }
    }
  }
  f = fopen(argv[4], "wb");
  if (!f) die("cannot open output");
  fwrite(L[output].y, 4, (size_t)L[output].out_dim, f);
  fclose(f);
  return 0;
}
