// tests/emu/selftest -- kernels with known answers for the emulator itself (tests/test_emu_selfcheck.py): cross-lane operations, DPP
// controls, the MFMA lane layout, workgroup barriers, co-resident workgroups behind a device-wide barrier, host <-> device flags.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DPP_ADD(v, ctrl, rmask, bound) \
  (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, (bound)))

__global__ void k_lane_ops(const float *__restrict__ in, float *__restrict__ out_xor, float *__restrict__ out_dpp, unsigned long long *__restrict__ ballots,
                           int *__restrict__ first, float *__restrict__ up8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = in[i];
  float s = v;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);  // every lane: the wave's sum (butterfly order)
  out_xor[i] = s;
  float a = v;   // the DPP scan of csrc/sos_tracker.hip: lane 63 holds the wave's sum
  DPP_ADD(a, 0x111, 0xf, true);
  DPP_ADD(a, 0x112, 0xf, true);
  DPP_ADD(a, 0x114, 0xf, true);
  DPP_ADD(a, 0x118, 0xf, true);
  DPP_ADD(a, 0x142, 0xa, false);
  DPP_ADD(a, 0x143, 0xc, false);
  out_dpp[i] = a;
  const unsigned long long b = __ballot(v > 0.f);
  if ((threadIdx.x & 63) == 0) ballots[i >> 6] = b;
  // a divergent region: only the lanes with a positive value take part; readfirstlane = the lowest of them
  if (v > 0.f) first[i] = __builtin_amdgcn_readfirstlane(i);
  else first[i] = -1;
  up8[i] = __shfl_up(v, 1, 8);
}

// D = A (16 x 4k) B (4k x 16) with v_mfma_f32_16x16x4_f32: A row-major [16][K], B row-major [K][16]
__global__ void k_mfma(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ D, int K) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, col = lane & 15;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[col * K + k0 + kq], B[(k0 + kq) * 16 + col], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(kq * 4 + r) * 16 + col] = acc[r];
}

// D = A (16 x 4k) B (4k x 16) with v_mfma_f64_16x16x4_f64: operands as the f32 form, ACCUMULATOR rows (lane >> 4) + 4 * reg
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma_f64(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ D, int K) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, col = lane & 15;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[col * K + k0 + kq], B[(k0 + kq) * 16 + col], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(kq + 4 * r) * 16 + col] = acc[r];
}
// v_mov_b64_dpp row_newbcast:n -- lane n of every 16-lane row to the whole row
template <int N>
__device__ double bc64(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xf, 0xf, true); }
__global__ void k_row_newbcast(const double *__restrict__ in, double *__restrict__ out) {
  const double v = in[threadIdx.x];
  out[threadIdx.x] = bc64<0>(v);
  out[64 + threadIdx.x] = bc64<5>(v);
  out[128 + threadIdx.x] = bc64<15>(v);
}
__global__ void k_lds_hog(float *out) {
  extern __shared__ float big[];
  big[threadIdx.x] = 1.f;
  __syncthreads();
  out[threadIdx.x] = big[threadIdx.x];
}

// barriers and divergence.  mode 0: a barrier in both arms of `tid < n` (n = 112: wave 1 has lanes on both sides) -- the shape of round
// 5's k_gn_solve, which a GPU runs one barrier out of step and the emulator must REFUSE; mode 1: a guarded barrier some lanes of wave
// 1 skip on their way to the next one -- refused too; mode 2: the same guard at wave granularity (n = 128: whole waves skip or
// take it, an anonymous s_barrier pairs whatever the waves execute) and mode 3: the barrier hoisted out of the guard -- both accepted.
__global__ void k_barrier_shapes(float *out, int n, int mode) {
  __shared__ float s[256];
  const int tid = threadIdx.x;
  s[tid] = (float)tid;
  __syncthreads();
  float v = 0.f;
  if (mode == 0) {
    if (tid < n) { v = s[(tid + 1) % n]; __syncthreads(); s[tid] = v; } else { __syncthreads(); }
  } else if (mode == 1 || mode == 2) {
    if (tid < n) { v = s[(tid + 1) % n]; __syncthreads(); s[tid] = v; }
    if (mode == 2 && tid >= n) __syncthreads();
  } else {
    if (tid < n) v = s[(tid + 1) % n];
    __syncthreads();
    if (tid < n) s[tid] = v;
  }
  __syncthreads();
  out[tid] = s[tid];
}

// LDS transpose of a 32 x 32 tile by 256 threads (4 waves), one barrier
__global__ void k_transpose(const float *__restrict__ in, float *__restrict__ out) {
  __shared__ float tile[32][33];
  const int t = threadIdx.x;
  for (int q = t; q < 1024; q += 256) tile[q >> 5][q & 31] = in[blockIdx.x * 1024 + q];
  __syncthreads();
  for (int q = t; q < 1024; q += 256) out[blockIdx.x * 1024 + q] = tile[q & 31][q >> 5];
  const int c = __syncthreads_count(t < 100);
  if (t == 0) out[blockIdx.x * 1024] += 0.f * c;
}

// every workgroup writes its slot, meets the others at a device-wide barrier, then sums what ALL of them wrote
__global__ void k_grid_barrier(float *__restrict__ slots, float *__restrict__ sums, unsigned *ctr) {
  if (threadIdx.x == 0) slots[blockIdx.x] = (float)(blockIdx.x + 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0;
    for (unsigned b = 0; b < gridDim.x; b++) s += slots[b];
    sums[blockIdx.x] = s;
  }
}

// waits for the host's flag in mapped memory, answers through another one
__global__ void k_mailbox(const int *go, int *done, const double *x, double *y) {
  if (threadIdx.x == 0)
    while (__hip_atomic_load(go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 7) __builtin_amdgcn_s_sleep(4);
  __syncthreads();
  y[threadIdx.x] = 2.0 * x[threadIdx.x];
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(done, 9, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int emu_selftest_lane_ops(const float *in, int n, float *out_xor, float *out_dpp, unsigned long long *ballots, int *first, float *up8) {
  float *d_in, *d_x, *d_d, *d_u;
  unsigned long long *d_b;
  int *d_f;
  hipMalloc(&d_in, 4 * n); hipMalloc(&d_x, 4 * n); hipMalloc(&d_d, 4 * n); hipMalloc(&d_u, 4 * n); hipMalloc(&d_b, 8 * (n / 64)); hipMalloc(&d_f, 4 * n);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipMemcpyAsync(d_in, in, 4 * n, hipMemcpyHostToDevice, st);
  k_lane_ops<<<n / 128, 128, 0, st>>>(d_in, d_x, d_d, d_b, d_f, d_u);
  hipMemcpyAsync(out_xor, d_x, 4 * n, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(out_dpp, d_d, 4 * n, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(ballots, d_b, 8 * (n / 64), hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(first, d_f, 4 * n, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(up8, d_u, 4 * n, hipMemcpyDeviceToHost, st);
  hipStreamSynchronize(st);
  hipFree(d_in); hipFree(d_x); hipFree(d_d); hipFree(d_u); hipFree(d_b); hipFree(d_f);
  hipStreamDestroy(st);
  return 0;
}
extern "C" int emu_selftest_mfma(const float *A, const float *B, float *D, int K) {
  float *dA, *dB, *dD;
  hipMalloc(&dA, 4 * 16 * K); hipMalloc(&dB, 4 * 16 * K); hipMalloc(&dD, 4 * 256);
  hipMemcpy(dA, A, 4 * 16 * K, hipMemcpyHostToDevice);
  hipMemcpy(dB, B, 4 * 16 * K, hipMemcpyHostToDevice);
  k_mfma<<<1, 64, 0, 0>>>(dA, dB, dD, K);
  hipDeviceSynchronize();
  hipMemcpy(D, dD, 4 * 256, hipMemcpyDeviceToHost);
  hipFree(dA); hipFree(dB); hipFree(dD);
  return 0;
}
extern "C" int emu_selftest_mfma_f64(const double *A, const double *B, double *D, int K) {
  double *dA, *dB, *dD;
  hipMalloc(&dA, 8 * 16 * K); hipMalloc(&dB, 8 * 16 * K); hipMalloc(&dD, 8 * 256);
  hipMemcpy(dA, A, 8 * 16 * K, hipMemcpyHostToDevice);
  hipMemcpy(dB, B, 8 * 16 * K, hipMemcpyHostToDevice);
  k_mfma_f64<<<1, 64, 0, 0>>>(dA, dB, dD, K);
  hipDeviceSynchronize();
  hipMemcpy(D, dD, 8 * 256, hipMemcpyDeviceToHost);
  hipFree(dA); hipFree(dB); hipFree(dD);
  return 0;
}
extern "C" int emu_selftest_row_newbcast(const double *in, double *out) {
  double *di, *dout;
  hipMalloc(&di, 8 * 64); hipMalloc(&dout, 8 * 192);
  hipMemcpy(di, in, 8 * 64, hipMemcpyHostToDevice);
  k_row_newbcast<<<1, 64, 0, 0>>>(di, dout);
  hipDeviceSynchronize();
  hipMemcpy(out, dout, 8 * 192, hipMemcpyDeviceToHost);
  hipFree(di); hipFree(dout);
  return 0;
}
// a launch that asks for more LDS than a compute unit has (160 KB on gfx950) must not be emulated as if it fitted
extern "C" int emu_selftest_lds_limit(int kbytes) {
  float *dout;
  hipMalloc(&dout, 4 * 64);
  k_lds_hog<<<1, 64, (size_t)kbytes * 1024, 0>>>(dout);
  hipDeviceSynchronize();
  hipFree(dout);
  return 0;
}
extern "C" int emu_selftest_barrier_shapes(float *out, int n, int mode) {
  float *dout;
  hipMalloc(&dout, 4 * 256);
  k_barrier_shapes<<<1, 256, 0, 0>>>(dout, n, mode);
  hipDeviceSynchronize();
  hipMemcpy(out, dout, 4 * 256, hipMemcpyDeviceToHost);
  hipFree(dout);
  return 0;
}
extern "C" int emu_selftest_transpose(const float *in, float *out, int nblocks) {
  float *di, *dout;
  hipMalloc(&di, 4096 * nblocks); hipMalloc(&dout, 4096 * nblocks);
  hipMemcpy(di, in, 4096 * nblocks, hipMemcpyHostToDevice);
  k_transpose<<<nblocks, 256, 0, 0>>>(di, dout);
  hipDeviceSynchronize();
  hipMemcpy(out, dout, 4096 * nblocks, hipMemcpyDeviceToHost);
  hipFree(di); hipFree(dout);
  return 0;
}
extern "C" int emu_selftest_grid_barrier(int nblocks, float *sums) {
  float *slots, *dsums;
  unsigned *ctr;
  hipMalloc(&slots, 4 * nblocks); hipMalloc(&dsums, 4 * nblocks); hipMalloc(&ctr, 4);
  hipMemset(ctr, 0, 4);
  k_grid_barrier<<<nblocks, 64, 0, 0>>>(slots, dsums, ctr);
  hipDeviceSynchronize();
  hipMemcpy(sums, dsums, 4 * nblocks, hipMemcpyDeviceToHost);
  hipFree(slots); hipFree(dsums); hipFree(ctr);
  return 0;
}
extern "C" int emu_selftest_mailbox(const double *x, double *y, int delay_us) {
  char *pin;
  hipHostMalloc(&pin, 4096, hipHostMallocMapped);
  int *go = reinterpret_cast<int *>(pin), *done = go + 1;
  double *px = reinterpret_cast<double *>(pin + 64), *py = px + 64;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  k_mailbox<<<1, 64, 0, st>>>(go, done, px, py);   // enqueued before the data exists
  for (volatile int spin = 0; spin < delay_us * 100; spin++) {}
  for (int i = 0; i < 64; i++) px[i] = x[i];
  __atomic_store_n(go, 7, __ATOMIC_RELEASE);
  while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != 9) {}
  for (int i = 0; i < 64; i++) y[i] = py[i];
  hipStreamSynchronize(st);
  hipStreamDestroy(st);
  hipHostFree(pin);
  return 0;
}
