/* TEST / MEASUREMENT INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).
 *
 * CPU baseline "B0" of BASELINE.md section 3: the same published semantics as oracle/xvec_oracle.c (what Kaldi's
 * nnet3-xvector-compute computes for the graphs of egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:94-114 and the
 * c-vector variants), written for speed instead of for reading: per layer the spliced input is
 * gathered into a contiguous matrix and multiplied with a register-blocked fp32 GEMM (the compiler vectorises the
 * inner loops: build with -O3 -march=<host>), utterances are spread over OpenMP threads - the way the reference spreads
 * them over `nj` processes (extract_xvectors_new.sh:91-93) - and the job goes file in, file out like the real tool:
 *
 *   xvec_cpu_baseline <program.bin> <feats.ark> <out.ark> [threads]
 *
 * program.bin: oracle/export_program.py; feats.ark: binary Kaldi archive of float matrices ("FM "); out.ark: binary
 * archive of float vectors ("FV ").  One embedding per utterance (no chunking: --chunk-size=-1).  Prints one line
 *   xvec_cpu_baseline: <n> utterances, <frames> frames, threads <t>, read <s> s, compute <s> s, write <s> s
 * It is NOT Kaldi (which the reference does not vendor and this image cannot build), and it is never linked into or
 * called by the product; bench.py times it as `cpu_baseline` and tests/test_oracle_c.py checks it against the numpy oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_SRC 8
/* register block of the GEMM micro-kernel: MR rows x NR columns of accumulators held in vector registers
 * (AVX-512: 12 x 32 floats = 24 of the 32 zmm; AVX2: 6 x 16 floats = 12 of the 16 ymm), written with GCC vector types */
#if defined(__AVX512F__)
#define VL 16
#define MR 12
#define NR 32
#else
#define VL 8
#define MR 6
#define NR 16
#endif
#define NV (NR / VL)
typedef float vf __attribute__((vector_size(VL * 4), aligned(4)));
typedef int vi __attribute__((vector_size(VL * 4), aligned(4)));

typedef struct {
  int32_t nsrc, src[MAX_SRC], off[MAX_SRC], dim[MAX_SRC];
  int32_t in_dim, out_dim, relu, bn, segment;
  float *wt; /* [in_dim][n_pad] (n_pad = out_dim rounded up to NR, zero filled) */
  float *bias, *scale, *offset;
  int n_pad, lo, right;
} layer_t;

static void die(const char* m) {
  fprintf(stderr, "xvec_cpu_baseline: %s\n", m);
  exit(2);
}
static void rd(void* p, size_t n, FILE* f) {
  if (fread(p, 1, n, f) != n) die("short read");
}
static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* C[m][n_pad] = A[m][k] . B[k][n_pad] + bias, then ReLU / scale-offset.
 * Ap: A packed in blocks of MR rows, each block [k][MR] (the MR values of one k are contiguous: broadcast loads
 *     walk one cache line stream); Bp: the current NR-column panel of B packed [k][NR].  Register block MR x NR. */
static void gemm_bias_act(const float* Ap, int m, int k, const layer_t* L, float* C, float* Bp) {
  const int np = L->n_pad;
  const float* B = L->wt;
  for (int n0 = 0; n0 < np; n0 += NR) {
    for (int kk = 0; kk < k; ++kk) memcpy(Bp + (size_t)kk * NR, B + (size_t)kk * np + n0, NR * 4);
    for (int i0 = 0; i0 < m; i0 += MR) {
      const int mr = m - i0 < MR ? m - i0 : MR;
      const float* a = Ap + (size_t)i0 * k;   /* block i0 / MR starts at row offset i0 * k */
      /* The accumulators are named variables, not an array: GCC keeps an array of vectors this large in memory and
       * stores all of it back in every pass of the k loop (16 stores per 16 FMAs in the first version of this kernel).
       * NV == 2 in both builds: row i owns (cia, cib). */
#define XV_ROWS(F) F(0) F(1) F(2) F(3) F(4) F(5) XV_ROWS_HI(F)
#if MR == 12
#define XV_ROWS_HI(F) F(6) F(7) F(8) F(9) F(10) F(11)
#else
#define XV_ROWS_HI(F)
#endif
#define XV_DECL(i) vf c##i##a = (vf){0}, c##i##b = (vf){0};
#define XV_FMA(i)                      \
  {                                    \
    const float x = a[(size_t)kk * MR + i]; \
    c##i##a += x * b0;                 \
    c##i##b += x * b1;                 \
  }
#define XV_PUT(i)    \
  acc[i][0] = c##i##a; \
  acc[i][1] = c##i##b;
      XV_ROWS(XV_DECL)
      for (int kk = 0; kk < k; ++kk) {
        const vf b0 = *(const vf*)(Bp + (size_t)kk * NR), b1 = *(const vf*)(Bp + (size_t)kk * NR + VL);
        XV_ROWS(XV_FMA)
      }
      vf acc[MR][NV];
      XV_ROWS(XV_PUT)
      for (int v = 0; v < NV; ++v) {
        const vf bias = *(const vf*)(L->bias + n0 + v * VL);
        const vf zero = (vf){0};
        vf sc = zero + 1.0f, of = zero;
        if (L->bn) {
          sc = *(const vf*)(L->scale + n0 + v * VL);
          of = *(const vf*)(L->offset + n0 + v * VL);
        }
        for (int i = 0; i < MR; ++i) {
          vf x = acc[i][v] + bias;
          if (L->relu) x = (vf)((vi)x & (vi)(x > zero));   /* lanes that are not > 0 become +0 */
          if (L->bn) x = x * sc + of;
          if (i < mr) *(vf*)(C + (size_t)(i0 + i) * np + n0 + v * VL) = x;
        }
      }
    }
  }
}

typedef struct {
  char key[64];
  int rows, cols;
  float* data;
} utt_t;

int main(int argc, char** argv) {
  if (argc < 4) die("usage: xvec_cpu_baseline <program.bin> <feats.ark> <out.ark> [threads]");
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
    const size_t NP = (N + NR - 1) / NR * NR;
    L[l].n_pad = (int)NP;
    float* w = (float*)malloc(K * N * 4);
    rd(w, K * N * 4, f); /* [N][K] row-major, Kaldi <LinearParams> orientation */
    L[l].wt = (float*)calloc(K * NP, 4);
    for (size_t n = 0; n < N; ++n)
      for (size_t k = 0; k < K; ++k) L[l].wt[k * NP + n] = w[n * K + k];
    free(w);
    L[l].bias = (float*)calloc(NP, 4);
    L[l].scale = (float*)calloc(NP, 4);
    L[l].offset = (float*)calloc(NP, 4);
    rd(L[l].bias, N * 4, f);
    rd(L[l].scale, N * 4, f);
    rd(L[l].offset, N * 4, f);
    if (!L[l].segment) {
      for (int j = 0; j < L[l].nsrc; ++j) {
        const int s = L[l].src[j];
        const int sl = s < 0 ? 0 : L[s].lo, sr = s < 0 ? 0 : L[s].right;
        if (sl - L[l].off[j] > L[l].lo) L[l].lo = sl - L[l].off[j];
        if (sr + L[l].off[j] > L[l].right) L[l].right = sr + L[l].off[j];
      }
    }
  }
  fclose(f);

  /* ---- read the whole archive (the sample is bounded by the caller) */
  const double t_read0 = now();
  f = fopen(argv[2], "rb");
  if (!f) die("cannot open feature archive");
  size_t n_utt = 0, cap = 0;
  utt_t* U = NULL;
  long total_frames = 0;
  for (;;) {
    utt_t u;
    int c, kl = 0;
    while ((c = fgetc(f)) != EOF && c != ' ')
      if (kl < 63) u.key[kl++] = (char)c;
    if (c == EOF) break;
    u.key[kl] = 0;
    char hdr[5];
    rd(hdr, 2, f);
    if (hdr[0] != 0 || hdr[1] != 'B') die("feature archive is not binary");
    rd(hdr, 3, f);
    if (memcmp(hdr, "FM ", 3)) die("only float matrices (FM) are supported");
    uint8_t sz;
    int32_t rows, cols;
    rd(&sz, 1, f);
    rd(&rows, 4, f);
    rd(&sz, 1, f);
    rd(&cols, 4, f);
    if (cols != input_dim) die("feature dimension does not match the model");
    u.rows = rows;
    u.cols = cols;
    u.data = (float*)malloc((size_t)rows * cols * 4);
    rd(u.data, (size_t)rows * cols * 4, f);
    if (n_utt == cap) {
      cap = cap ? cap * 2 : 1024;
      U = (utt_t*)realloc(U, cap * sizeof(utt_t));
    }
    U[n_utt++] = u;
    total_frames += rows;
  }
  fclose(f);
  const double t_read = now() - t_read0;

  int threads = argc > 4 ? atoi(argv[4]) : 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  threads = omp_get_max_threads();
#else
  threads = 1;
#endif
  const int E = L[output].out_dim;
  float* emb = (float*)calloc(n_utt * (size_t)E, 4);
  int* okv = (int*)calloc(n_utt, sizeof(int));

  const double t_c0 = now();
#pragma omp parallel
  {
    float** Y = (float**)calloc((size_t)n_layers, sizeof(float*)); /* this thread's activations, grown on demand */
    size_t* Ycap = (size_t*)calloc((size_t)n_layers, sizeof(size_t));
    float* X = NULL; /* spliced input matrix, packed in blocks of MR rows: block b = [K][MR] */
    size_t Xcap = 0, Bcap = 0;
    float* Bp = NULL;
    float* stats = NULL;
#pragma omp for schedule(dynamic, 1)
    for (long ui = 0; ui < (long)n_utt; ++ui) {
      const utt_t* u = &U[ui];
      const int T = u->rows;
      int fail = 0;
      for (int l = 0; l < n_layers && !fail; ++l) {
        const layer_t* a = &L[l];
        const int np = a->n_pad, K = a->in_dim;
        const int m = a->segment ? 1 : T - a->lo - a->right;
        if (m < 1) {
          fail = 1; /* nnet3 never pads: the utterance counts as failed */
          break;
        }
        if ((size_t)m * np > Ycap[l]) {
          Ycap[l] = (size_t)m * np * 2;
          Y[l] = (float*)realloc(Y[l], Ycap[l] * 4);
        }
        const int mp = (m + MR - 1) / MR * MR;
        if ((size_t)mp * K > Xcap) {
          Xcap = (size_t)mp * K * 2;
          X = (float*)realloc(X, Xcap * 4);
        }
        if ((size_t)K * NR > Bcap) {
          Bcap = (size_t)K * NR;
          Bp = (float*)realloc(Bp, Bcap * 4);
        }
        if (mp != m) memset(X + (size_t)(mp - MR) * K, 0, (size_t)MR * K * 4); /* rows of the last block beyond m */
        /* gather: row i of X = concat_j source_j[t + off_j], t = lo + i */
        int k0 = 0;
        for (int j = 0; j < a->nsrc; ++j) {
          const int s = a->src[j], d = a->dim[j];
          for (int i = 0; i < m; ++i) {
            const float* srow;
            if (a->segment) srow = s == -2 ? stats : Y[s];
            else if (s < 0) srow = u->data + (size_t)(a->lo + i + a->off[j]) * input_dim;
            else srow = Y[s] + (size_t)(a->lo + i + a->off[j] - L[s].lo) * L[s].n_pad;
            float* dst = X + (size_t)(i / MR) * MR * K + (size_t)k0 * MR + (i % MR);
            for (int q = 0; q < d; ++q) dst[(size_t)q * MR] = srow[q];
          }
          k0 += d;
        }
        gemm_bias_act(X, m, K, a, Y[l], Bp);
        if (l == pooled) {
          const int N = a->out_dim;
          stats = (float*)realloc(stats, (size_t)2 * N * 4);
          for (int n = 0; n < N; ++n) stats[n] = stats[N + n] = 0.f;
          for (int i = 0; i < m; ++i) {
            const float* y = Y[l] + (size_t)i * np;
            for (int n = 0; n < N; ++n) {
              stats[n] += y[n];
              stats[N + n] += y[n] * y[n];
            }
          }
          for (int n = 0; n < N; ++n) {
            const float mu = stats[n] / (float)m;
            volatile float m2 = mu * mu; /* separately rounded, like Kaldi's AddVecVec */
            float var = stats[N + n] / (float)m - m2;
            if (var < var_floor) var = var_floor;
            stats[n] = mu;
            stats[N + n] = sqrtf(var);
          }
        }
      }
      if (!fail) {
        memcpy(emb + (size_t)ui * E, Y[output], (size_t)E * 4);
        okv[ui] = 1;
      }
    }
    for (int l = 0; l < n_layers; ++l) free(Y[l]);
    free(Y);
    free(Ycap);
    free(X);
    free(Bp);
    free(stats);
  }
  const double t_compute = now() - t_c0;

  const double t_w0 = now();
  f = fopen(argv[3], "wb");
  if (!f) die("cannot open output archive");
  long n_ok = 0;
  for (size_t ui = 0; ui < n_utt; ++ui) {
    if (!okv[ui]) continue;
    fputs(U[ui].key, f);
    fputc(' ', f);
    fputc(0, f);
    fputc('B', f);
    fwrite("FV ", 1, 3, f);
    fputc(4, f);
    const int32_t dim = E;
    fwrite(&dim, 4, 1, f);
    fwrite(emb + ui * (size_t)E, 4, (size_t)E, f);
    ++n_ok;
  }
  fclose(f);
  const double t_write = now() - t_w0;
  printf("xvec_cpu_baseline: %ld utterances, %ld frames, threads %d, read %.3f s, compute %.3f s, write %.3f s\n", n_ok,
         total_frames, threads, t_read, t_compute, t_write);
  return n_ok > 0 ? 0 : 1;
}
