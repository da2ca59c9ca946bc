/*
 * Plain-C restatement of the fp8-weight (W8A16) fused_experts path — TEST INFRASTRUCTURE / CPU BASELINE.
 *
 * Follows the reference's embedded oracle, not any sgl_kernel source (which is absent):
 *   dequant        w[r][c] * scale[r/bn][c/bk] in fp32      /root/reference/test_moe_fp8_ext.py:22-25
 *   per (token,slot) row:  h = silu(x W1g^T) * (x W1u^T);  y = h W2^T   (fp32)      :70-86
 *   out[m] = sum_j topk_w[m][j] * y[m][j], ids outside [0,E) skipped (the -1 padding)
 *                                             :89-91, /root/reference/test_moe_offloading_cpu.py:33-52
 * All arithmetic fp32 (`omp simd` reductions; the compiler may contract a*b+c to fma), one thread per expert.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RB 16

static float e4m3fn_to_f32(uint8_t b) {
    /* OCP e4m3fn: 1-4-3, bias 7, no inf, 0x7f/0xff = NaN */
    int s = b >> 7, e = (b >> 3) & 0xF, m = b & 7;
    float v;
    if (e == 0xF && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);          /* subnormal: m/8 * 2^-6 */
    else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -v : v;
}

static float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int sglk_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* bench.py's cpu_baseline pins the run to one NUMA node (like /root/reference/run_bench_cpu.sh:14-20) and sets the thread count
 * to that node's cores */
void sglk_oracle_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

/* dequantise one expert matrix [R][C] fp8 with block scales [R/bn][C/bk] (ceil) into dst f32 */
static void dequant_matrix(const uint8_t* w, const float* scale, int R, int C, int bn, int bk,
                           const float* lut, float* dst) {
    int sc_cols = (C + bk - 1) / bk;
    for (int r = 0; r < R; ++r) {
        const float* srow = scale + (size_t)(r / bn) * sc_cols;
        const uint8_t* wr = w + (size_t)r * C;
        float* dr = dst + (size_t)r * C;
        for (int c = 0; c < C; ++c) dr[c] = lut[wr[c]] * srow[c / bk];
    }
}

static float dot_f32(const float* a, const float* b, int n) {
    float acc = 0.f;
#pragma omp simd reduction(+ : acc)
    for (int i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
}

/*
 * a        [M][K]   bf16 bits
 * w1       [E][2N][K] fp8 e4m3fn bytes, rows [0,N) = gate, [N,2N) = up
 * w2       [E][K][N]  fp8
 * w1_scale [E][2N/bn][K/bk] f32,  w2_scale [E][K/bn][N/bk] f32
 * topk_w   [M][topk] f32, topk_ids [M][topk] i32
 * out      [M][K] f32
 * returns 0, or -1 on allocation failure
 */
int sglk_oracle_fused_experts_fp8(const uint16_t* a, int M, int N, int K, int E, int topk,
                                  const uint8_t* w1, const uint8_t* w2, const float* w1_scale,
                                  const float* w2_scale, int bn, int bk, const float* topk_w,
                                  const int32_t* topk_ids, float* out) {
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = e4m3fn_to_f32((uint8_t)i);

    const size_t slots = (size_t)M * topk;
    float* x = (float*)malloc((size_t)M * K * sizeof(float));
    float* y = (float*)calloc(slots * K, sizeof(float)); /* per-slot expert output */
    int* order = (int*)malloc(slots * sizeof(int));      /* slots grouped by expert */
    int* start = (int*)calloc((size_t)E + 1, sizeof(int));
    if (!x || !y || !order || !start) return -1;

    for (size_t i = 0; i < (size_t)M * K; ++i) x[i] = bf16_to_f32(a[i]);

    for (size_t s = 0; s < slots; ++s) {
        int e = topk_ids[s];
        if (e >= 0 && e < E) start[e + 1]++;
    }
    for (int e = 0; e < E; ++e) start[e + 1] += start[e];
    {
        int* fill = (int*)malloc((size_t)E * sizeof(int));
        memcpy(fill, start, (size_t)E * sizeof(int));
        for (size_t s = 0; s < slots; ++s) {
            int e = topk_ids[s];
            if (e >= 0 && e < E) order[fill[e]++] = (int)s;
        }
        free(fill);
    }

    const int s1c = (K + bk - 1) / bk, s1r = (2 * N + bn - 1) / bn;
    const int s2c = (N + bk - 1) / bk, s2r = (K + bn - 1) / bn;
    int fail = 0;

#pragma omp parallel
    {
        float* W1 = (float*)malloc((size_t)2 * N * K * sizeof(float));
        float* W2 = (float*)malloc((size_t)K * N * sizeof(float));
        float* g = (float*)malloc((size_t)RB * 2 * N * sizeof(float));
        float* h = (float*)malloc((size_t)RB * N * sizeof(float));
        if (!W1 || !W2 || !g || !h) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int e = 0; e < E; ++e) {
                if (start[e + 1] == start[e]) continue;
                dequant_matrix(w1 + (size_t)e * 2 * N * K, w1_scale + (size_t)e * s1r * s1c, 2 * N, K, bn, bk, lut, W1);
                dequant_matrix(w2 + (size_t)e * K * N, w2_scale + (size_t)e * s2r * s2c, K, N, bn, bk, lut, W2);
                /* row blocks of RB so each dequantised weight row is reused from L1 across RB tokens */
                for (int p0 = start[e]; p0 < start[e + 1]; p0 += RB) {
                    const int nb = (start[e + 1] - p0 < RB) ? start[e + 1] - p0 : RB;
                    for (int n = 0; n < 2 * N; ++n)
                        for (int r = 0; r < nb; ++r)
                            g[(size_t)r * 2 * N + n] = dot_f32(x + (size_t)(order[p0 + r] / topk) * K, W1 + (size_t)n * K, K);
                    for (int r = 0; r < nb; ++r)
                        for (int n = 0; n < N; ++n) {
                            float gate = g[(size_t)r * 2 * N + n];
                            h[(size_t)r * N + n] = gate / (1.0f + expf(-gate)) * g[(size_t)r * 2 * N + N + n];
                        }
                    for (int k = 0; k < K; ++k)
                        for (int r = 0; r < nb; ++r)
                            y[(size_t)order[p0 + r] * K + k] = dot_f32(h + (size_t)r * N, W2 + (size_t)k * N, N);
                }
            }
        }
        free(W1); free(W2); free(g); free(h);
    }

    if (!fail) {
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m) {
            float* o = out + (size_t)m * K;
            for (int k = 0; k < K; ++k) o[k] = 0.f;
            for (int j = 0; j < topk; ++j) {
                int e = topk_ids[(size_t)m * topk + j];
                if (e < 0 || e >= E) continue;
                const float wgt = topk_w[(size_t)m * topk + j];
                const float* yr = y + ((size_t)m * topk + j) * K;
                for (int k = 0; k < K; ++k) o[k] += wgt * yr[k];
            }
        }
    }
    free(x); free(y); free(order); free(start);
    return fail ? -1 : 0;
}
