// Stand-alone attempt at the quarter-wave fault of DESIGN.md section 4 (round 4): one workgroup of 8 waves, waves 0-3 keep the
// matrix pipe busy (v_mfma_f32_16x16x32_f16 on register operands), waves 4-7 run the failing copy's arithmetic -- the
// LayerNorm affine of four values as packed-f32 operations, in the instruction order of the production listing -- on
// values loaded from memory, split the result into f16 hi / lo pairs, stage it in LDS (ds_write2_b64) and read it back
// after a barrier.  Every output is checked on the host against ((x - mean) * rstd) * gamma + beta.
//   hipcc --offload-arch=gfx950 -O3 tools/ln_hazard_micro.hip -o /tmp/ln_micro && /tmp/ln_micro [iterations]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void micro(const float* __restrict__ x, const float* __restrict__ stats,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ out, int rows, int planes, int iters, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave < 4) {
        // matrix pipe load: dependent chains of 16x16x32 MFMAs, one barrier per period like the compute waves
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        for (int it = 0; it < iters; ++it) {
            for (int p = 0; p < planes; p += 2) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                __syncthreads();
                // read the staged planes like the consumers would (keeps the LDS read side busy)
                f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane * 7 + p) % (2 * 128)) * 36);
                acc[0] += v;
            }
        }
        f32x4 s = acc[0];
        for (int i = 1; i < 8; ++i) s += acc[i];
        if (s[0] == 12345.678f) sink[tid] = s[1];
        return;
    }
    // loader waves: thread = (row lane, channel quad), four rows per thread and plane, two planes per barrier
    const int lt = tid - 256, c4 = lt & 7, arow = lt >> 3;
    const int cin = planes * 32;
    for (int it = 0; it < iters; ++it) {
        for (int p = 0; p < planes; p += 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int ch = (p + sub) * 32 + c4 * 4;
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + ch);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(beta + ch);
                f32x4 xv[4];
                f32x2 st[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = arow + 32 * j;
                    xv[j] = *reinterpret_cast<const f32x4*>(x + (long)row * cin + ch);
                    st[j] = *reinterpret_cast<const f32x2*>(stats + row * 2);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = xv[j];
                    v = (v - st[j].x) * st[j].y * g + bq;                     // (compiles to the v_sub x4 / v_pk_mul / v_pk_fma chain)
                    // split into f16 hi / lo and stage
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float t = v[e]; asm volatile("" : "+v"(t)); h[e] = (_Float16)t; l[e] = (_Float16)(t - (float)h[e]); }
                    _Float16* base = reinterpret_cast<_Float16*>(lds + (size_t)(sub * 128 + arow + 32 * j) * 36) + (c4 >> 1) * 16 + (c4 & 1) * 4;
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<f16x4*>(base) = f16x4{h[0], h[1], h[2], h[3]};
                    *reinterpret_cast<f16x4*>(base + 8) = f16x4{l[0], l[1], l[2], l[3]};
                }
            }
            __syncthreads();
            if (it == iters - 1) {
                // read back what was staged (after the barrier, like a consumer) and publish hi + lo
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                        const _Float16* base = reinterpret_cast<const _Float16*>(lds + (size_t)(sub * 128 + arow + 32 * j) * 36) + (c4 >> 1) * 16 + (c4 & 1) * 4;
                        const f16x4 h = *reinterpret_cast<const f16x4*>(base), l = *reinterpret_cast<const f16x4*>(base + 8);
                        f32x4 o;
                        for (int e = 0; e < 4; ++e) o[e] = (float)h[e] + (float)l[e];
                        *reinterpret_cast<f32x4*>(out + (long)(arow + 32 * j) * cin + (p + sub) * 32 + c4 * 4) = o;
                    }
            }
            // (the next period overwrites the slots only after the consumers' barrier of that period)
        }
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50, launches = argc > 2 ? atoi(argv[2]) : 200;
    const int rows = 128, planes = 16, cin = planes * 32;
    std::vector<float> x(rows * cin), st(rows * 2), g(cin), b(cin), ref(rows * cin), out(rows * cin);
    srand(7);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f) * 4.f; };
    for (auto& v : x) v = rnd();
    for (auto& v : g) v = rnd();
    for (auto& v : b) v = rnd();
    for (int r = 0; r < rows; ++r) {
        double s = 0, ss = 0;
        for (int c = 0; c < cin; ++c) { s += x[r * cin + c]; ss += (double)x[r * cin + c] * x[r * cin + c]; }
        const double mean = s / cin, var = ss / cin - mean * mean;
        st[r * 2] = (float)mean;
        st[r * 2 + 1] = (float)(1.0 / std::sqrt(var + 1e-5));
        for (int c = 0; c < cin; ++c) ref[r * cin + c] = ((x[r * cin + c] - st[r * 2]) * st[r * 2 + 1]) * g[c] + b[c];
    }
    float *dx, *ds, *dg, *db, *dout, *dsink;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&ds, st.size() * 4); hipMalloc(&dg, g.size() * 4); hipMalloc(&db, b.size() * 4);
    hipMalloc(&dout, out.size() * 4); hipMalloc(&dsink, 512 * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, st.data(), st.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    long bad_total = 0;
    int bad_launches = 0, hist[8] = {0};
    for (int L = 0; L < launches; ++L) {
        hipMemset(dout, 0xff, out.size() * 4);
        hipLaunchKernelGGL(micro, dim3(1), dim3(512), 0, 0, dx, ds, dg, db, dout, rows, planes, iters, dsink);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cin; ++c) {
                const float d = std::fabs(out[r * cin + c] - ref[r * cin + c]);
                if (!(d <= 2e-5f * std::fmax(1.f, std::fabs(ref[r * cin + c])))) {
                    if (bad_total + bad < 8)
                        printf("  launch %d row %d channel %d: got %.7g expected %.7g beta %.7g\n", L, r, c, out[r * cin + c], ref[r * cin + c], b[c]);
                    ++bad; ++hist[r & 7];
                }
            }
        bad_total += bad;
        bad_launches += bad > 0;
    }
    printf("iterations %d, launches %d: %d launches with wrong values, %ld wrong values, rows mod 8 [%d %d %d %d %d %d %d %d]\n", iters, launches,
           bad_launches, bad_total, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    return 0;
}
