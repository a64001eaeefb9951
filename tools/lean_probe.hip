// Issue / throughput rates the lean deformable-conv sampler depends on, one sampler-like wave per SIMD, alone and beside a wave of the
// same SIMD that issues v_mfma_f32_32x32x2_f32 back to back.  Prints core clocks per instruction (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/lean_probe.hip -o /tmp/lean_probe && /tmp/lean_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode: 0 v_fma_f32, 1 v_pk_fma_f32, 2 ds_read_b128 contiguous cells, 3 ds_read_b128 random cells, 4 ds_read_b128 jittered cells (lane + small random),
//       5 the sampler's mix (8 reads + 16 pk_fma per group, next group's reads in flight), jittered cells
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters, int with_mfma, const int* cells) {
  extern __shared__ __align__(16) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 25600; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
  __syncthreads();
  const bool mf = wave >= 4;          // waves 4-7: second resident wave of each SIMD
  float s = 0.f;
  long long t0 = 0, t1 = 0;
  if (mf) {
    if (with_mfma == 1 || with_mfma == 3) {
      f32x16 a0 = {0}, a1 = {0};
      float fa = 1.0001f * lane, fb = 0.5f;
      for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, a1, 0, 0, 0);
        }
      }
      s = a0[0] + a1[3];
    } else if (with_mfma == 5) {
      f32x16 a0 = {0}, a1 = {0};
      f32x4 fa = {1.f, 2.f, 3.f, 4.f}, fb = {0.5f, 0.25f, 0.125f, 1.f};
      for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(fa), "v"(fb));
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(fb), "v"(fa));
        }
      }
      s = a0[0] + a1[3];
    } else if (with_mfma == 2 || with_mfma == 4) {
      f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      float fa = 1.0001f * lane, fb = 0.5f;
      for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, a3, 0, 0, 0);
        }
      }
      s = a0[0] + a1[3] + a2[1] + a3[2];
    }
  } else {
    f32x2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)i, 1.f};
    f32x2 w = {1.0001f, 0.9999f}, v = {0.5f, 0.25f};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = i;
    if (with_mfma == 3 || with_mfma == 4) __builtin_amdgcn_s_setprio(3);
    const int cell = cells[(blockIdx.x * 4 + wave) * 64 + lane];
    const unsigned base = (unsigned)((MODE == 2 ? lane : cell) * 16);   // dynamic LDS starts at address 0
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[j & 7]) : "v"(w.x), "v"(v.x));
      } else if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j & 7]) : "v"(w), "v"(v));
      } else if (MODE <= 4) {
        f32x4 r[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(base), "n"(j * 640 + g * 16));
          asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
          for (int j = 0; j < 8; ++j) s += r[j].x;
        }
      } else {
        f32x4 r[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[0][j]) : "v"(base), "n"(j * 640));
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (g + 1 < 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[(g + 1) & 1][j]) : "v"(base), "n"(j * 640 + 16));
            asm volatile("s_waitcnt lgkmcnt(8)");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)");
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f32x2 lo = {r[g & 1][j].x, r[g & 1][j].y}, hi = {r[g & 1][j].z, r[g & 1][j].w};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0]) : "v"(w), "v"(lo));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[1]) : "v"(w), "v"(hi));
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15");
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char* what, int per_iter, int with_mfma, float* out, long long* clk, const int* cells) {
  const int iters = 500;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024);
  float ms = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 110 * 1024, 0, out, clk, iters, with_mfma, cells);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  static const char* names[] = {"alone", "beside 32x32x2", "beside 16x16x4", "beside 32x32x2, VALU wave prio 3", "beside 16x16x4, VALU wave prio 3", "beside 32x32x16 bf16"};
  printf("%-58s %-34s %7.2f clk/instr  (%.3f ms)\n", what, names[with_mfma], (double)c / iters / per_iter, ms);
}

int main() {
  float* out; long long* clk; int *c_rand, *c_jit;
  hipMalloc(&out, 4 * 512 * 256); hipMalloc(&clk, 8); hipMalloc(&c_rand, 4 * 256 * 4 * 64); hipMalloc(&c_jit, 4 * 256 * 4 * 64);
  static int h_rand[256 * 4 * 64], h_jit[256 * 4 * 64];
  unsigned st = 12345;
  for (int i = 0; i < 256 * 4 * 64; ++i) {
    st = st * 1664525u + 1013904223u; h_rand[i] = (st >> 8) % 1500;
    st = st * 1664525u + 1013904223u; h_jit[i] = 100 + (i & 63) + (int)((st >> 8) % 5) - 2 + 40 * (int)((st >> 20) % 3);   // lane + jitter(-2..2) + row jitter
  }
  hipMemcpy(c_rand, h_rand, sizeof(h_rand), hipMemcpyHostToDevice);
  hipMemcpy(c_jit, h_jit, sizeof(h_jit), hipMemcpyHostToDevice);
  for (int mf = 0; mf < 6; mf += 5) {
    run<0>("v_fma_f32 x64 (8 chains)", 64, mf, out, clk, c_rand);
    run<1>("v_pk_fma_f32 x64 (8 chains)", 64, mf, out, clk, c_rand);
    run<2>("ds_read_b128 x64 contiguous cells, 8 in flight", 64, mf, out, clk, c_rand);
    run<3>("ds_read_b128 x64 random cells, 8 in flight", 64, mf, out, clk, c_rand);
    run<4>("ds_read_b128 x64 jittered cells, 8 in flight", 64, mf, out, clk, c_jit);
    run<5>("sampler mix: 64 reads + 128 pk_fma, jittered (per instr of 192)", 192, mf, out, clk, c_jit);
  }
  return 0;
}
