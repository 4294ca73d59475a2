// tests/emu -- TEST INFRASTRUCTURE, not product code.
//
// A lockstep SIMT emulator that lets the UNMODIFIED kernels of sos_slam_amd/csrc/*.hip execute on the host CPU while no MI355X
// is reachable: this header stands in for <hip/hip_runtime.h> when tests/emu/build_emu.py compiles the (textually preprocessed)
// sources with the host clang++.  Every GPU thread is a fiber; the 64 lanes of a wavefront meet at every cross-lane operation
// (__shfl*, __ballot, readlane, DPP, MFMA) and the threads of a workgroup at __syncthreads, so the kernels' data flow -- indexing,
// LDS staging, reductions trees, flags, host/device hand-shakes -- is executed exactly as written.  It proves nothing about
// performance, occupancy, hardware rounding of rcp/rsq/MFMA or memory-model races; a run under it is NOT a GPU run and is never
// reported as one (see tests/emu/README.md).  Nothing under sos_slam_amd/ refers to this directory.
#pragma once
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

// ------------------------------------------------------------------------------------------------ qualifiers
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))

// ------------------------------------------------------------------------------------------------ vector types
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
#define EMU_VEC2(T, N) struct alignas(2 * sizeof(T)) N { T x, y; }; static inline N make_##N(T x, T y) { return N{x, y}; }
#define EMU_VEC3(T, N) struct N { T x, y, z; }; static inline N make_##N(T x, T y, T z) { return N{x, y, z}; }
#define EMU_VEC4(T, N) struct alignas(4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T)) N { T x, y, z, w; }; static inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
EMU_VEC2(float, float2) EMU_VEC3(float, float3) EMU_VEC4(float, float4)
EMU_VEC2(double, double2) EMU_VEC3(double, double3) EMU_VEC4(double, double4)
EMU_VEC2(int, int2) EMU_VEC3(int, int3) EMU_VEC4(int, int4)
EMU_VEC2(unsigned, uint2) EMU_VEC4(unsigned, uint4)
EMU_VEC2(short, short2) EMU_VEC4(short, short4)
EMU_VEC2(unsigned short, ushort2) EMU_VEC4(unsigned short, ushort4)
EMU_VEC2(unsigned char, uchar2) EMU_VEC4(unsigned char, uchar4)
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{x, y, z}; }

// ------------------------------------------------------------------------------------------------ host API
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum : unsigned { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };
enum : unsigned { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63, hipDeviceAttributeMaxSharedMemoryPerBlock = 74 };
namespace emu { struct Stream; struct Event; }
typedef emu::Stream *hipStream_t;
typedef emu::Event *hipEvent_t;
struct hipDeviceProp_t {
  char name[256];
  size_t totalGlobalMem, sharedMemPerBlock;
  int multiProcessorCount, wavefrontWidth, maxThreadsPerBlock, clockRate, major, minor;
  char gcnArchName[256];
};

extern "C" {
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t emu_hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t emu_hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t emu_hipHostGetDevicePointer(void **d, void *h, unsigned flags);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipLaunchHostFunc(hipStream_t st, void (*fn)(void *), void *user);
hipError_t hipMemset(void *p, int v, size_t n);
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned flags);
hipError_t hipStreamCreate(hipStream_t *st);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamQuery(hipStream_t st);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
}
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return emu_hipMalloc((void **)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned flags = 0) { return emu_hipHostMalloc((void **)p, n, flags); }
template <class T> static inline hipError_t hipHostGetDevicePointer(T **d, void *h, unsigned flags) { return emu_hipHostGetDevicePointer((void **)d, h, flags); }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, F, int blockSize, size_t) {
  *n = blockSize >= 512 ? 1 : 2;
  return hipSuccess;
}
template <class T> static inline hipError_t hipMemcpyFromSymbol(void *dst, const T &sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
  hipDeviceSynchronize();
  memcpy(dst, (const char *)&sym + off, n);
  return hipSuccess;
}
template <class T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
  hipDeviceSynchronize();
  memcpy((char *)&sym + off, src, n);
  return hipSuccess;
}
#define HIP_SYMBOL(x) x

// ------------------------------------------------------------------------------------------------ the SIMT machine
namespace emu {
enum St : uint8_t { RUNNABLE = 0, WAIT_WAVE, WAIT_BLOCK, YIELDED, DEAD };
enum Op : int { OP_SHFL = 1, OP_SHFL_XOR, OP_SHFL_UP, OP_SHFL_DOWN, OP_BALLOT, OP_READFIRST, OP_READLANE, OP_DPP, OP_MFMA_16x16x4_F32, OP_WAVE_BARRIER, OP_SEQSUM8, OP_DPP64, OP_MFMA_16x16x4_F64 };
struct Block;
struct Fiber {
  void *sp;            // saved stack pointer while switched out
  uint3 tid;           // threadIdx
  Block *blk;
  int lin, lane, wave; // linear thread id in the block, lane in the wave, wave in the block
  St st;
  // a pending cross-lane operation: inputs, then outputs written by the resolver
  int op, p0, p1, p2, p3;
  const void *site;
  uint64_t in64, out64;
  float fin[6], fout[4];
  double din[6], dout[4];  // v_mfma_f64_16x16x4_f64: a, b, c[4] -> d[4]
  uint64_t old64;          // `old` operand of a 64-bit DPP move
  int pred;            // __syncthreads_or / _count
  int bar_id, bar_line;  // the __syncthreads statement this lane waits in
  const char *bar_file;
};
struct Block {
  dim3 idx, bdim, gdim;
  int nthreads, nwaves, alive, waiting_block;
  Fiber *fibers;
  void *dyn_shared;
  size_t dyn_bytes;
  size_t static_bytes = 0;          // __shared__ declarations reached so far
  int sync_acc_or, sync_acc_count;  // accumulated by the releasing barrier
  int sync_res_or, sync_res_count;
  void *statics;                    // per-block map: declaration key -> storage
};
extern thread_local Fiber *cur;
void wave_op();      // block the calling lane until its group has been resolved (noinline: the return address names the site)
void block_sync(int site_id, const char *file, int line);   // __syncthreads, named by its place in the source
void yield_lane();   // s_sleep: stay runnable, let everything else run first
void *shared_static_lookup(const void *key, size_t bytes, size_t align);
[[noreturn]] void die(const char *msg);

template <class T> static inline T *shared_static(const void *key) { return reinterpret_cast<T *>(shared_static_lookup(key, sizeof(T), alignof(T) > 16 ? alignof(T) : 16)); }
static inline void *dyn_shared() { return cur->blk->dyn_shared; }

struct LaunchBase {
  dim3 grid, block;
  size_t shmem;
  const char *name;
  virtual void run_thread() = 0;  // the kernel body for the calling fiber
  virtual ~LaunchBase() {}
};
void enqueue_launch(hipStream_t st, LaunchBase *l);

// the kernel is called BY NAME inside a generic lambda (default arguments and implicit conversions behave as at a <<< >>> site);
// the arguments are evaluated and copied when the launch is enqueued, as hipLaunchKernel copies the argument block
template <class F, class... A> struct Launch : LaunchBase {
  F f;
  std::tuple<std::decay_t<A>...> args;
  template <class... B> Launch(F f_, B &&...a) : f(f_), args(std::forward<B>(a)...) {}
  void run_thread() override { std::apply(f, args); }
};
template <class F, class... A> static inline void launch(const char *name, F f, dim3 g, dim3 b, size_t sh, hipStream_t st, A &&...a) {
  auto *l = new Launch<F, A...>(f, std::forward<A>(a)...);
  l->grid = g; l->block = b; l->shmem = sh; l->name = name;
  enqueue_launch(st, l);
}

template <class T> static inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "cross-lane value wider than 64 bits");
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  return u;
}
template <class T> static inline T from_bits(uint64_t u) {
  T v;
  memcpy(&v, &u, sizeof(T));
  return v;
}
template <class T> static inline __attribute__((always_inline)) T xlane(int op, T v, int p0, int p1) {
  Fiber *f = cur;
  f->op = op; f->in64 = to_bits(v); f->p0 = p0; f->p1 = p1;
  wave_op();
  return from_bits<T>(cur->out64);
}
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->blk->idx)
#define blockDim (emu::cur->blk->bdim)
#define gridDim (emu::cur->blk->gdim)

// s_barrier counts WAVES, not threads: a wave executes each s_barrier instruction it reaches once, with whatever lanes are active.  The
// emulation runs the lanes as independent fibers, so it must notice by itself when the lanes of one wave wait in DIFFERENT barrier
// statements (both arms of a lane-divergent branch holding one; a guarded barrier some lanes skipped): on hardware that wave passes
// two barriers where its siblings pass one and runs out of step from there on (round 5's k_gn_solve).  Every textual occurrence gets a
// number (__COUNTER__); the release in emu_runtime.cpp refuses a wave whose waiting lanes carry different numbers.
static inline __attribute__((always_inline)) int emu_syncthreads_or(int p, int id, const char *file, int line) {
  emu::cur->pred = p;
  emu::block_sync(id, file, line);
  return emu::cur->blk->sync_res_or;
}
static inline __attribute__((always_inline)) int emu_syncthreads_count(int p, int id, const char *file, int line) {
  emu::cur->pred = p;
  emu::block_sync(id, file, line);
  return emu::cur->blk->sync_res_count;
}
#define __syncthreads() emu::block_sync(__COUNTER__, __FILE__, __LINE__)
#define __syncthreads_or(p) emu_syncthreads_or((p), __COUNTER__, __FILE__, __LINE__)
#define __syncthreads_count(p) emu_syncthreads_count((p), __COUNTER__, __FILE__, __LINE__)
// A wave executes in lockstep: every store a lane issues before a fence is visible before anything any lane of the wave does after
// it ("all lanes store, fence, lane 0 raises the flag").  Fibers run independently between meeting points, so the fences are made
// meeting points of the lanes that execute them.
static inline __attribute__((always_inline)) void __threadfence() {
  (void)emu::xlane<int>(emu::OP_WAVE_BARRIER, 0, 0, 0);
  std::atomic_thread_fence(std::memory_order_seq_cst);
}
static inline __attribute__((always_inline)) void __threadfence_system() { __threadfence(); }
static inline __attribute__((always_inline)) void __threadfence_block() { __threadfence(); }

template <class T> static inline __attribute__((always_inline)) T __shfl(T v, int src, int width = 64) { return emu::xlane(emu::OP_SHFL, v, src, width); }
template <class T> static inline __attribute__((always_inline)) T __shfl_xor(T v, int m, int width = 64) { return emu::xlane(emu::OP_SHFL_XOR, v, m, width); }
template <class T> static inline __attribute__((always_inline)) T __shfl_up(T v, unsigned d, int width = 64) { return emu::xlane(emu::OP_SHFL_UP, v, (int)d, width); }
template <class T> static inline __attribute__((always_inline)) T __shfl_down(T v, unsigned d, int width = 64) { return emu::xlane(emu::OP_SHFL_DOWN, v, (int)d, width); }
static inline __attribute__((always_inline)) unsigned long long __ballot(int p) { return emu::xlane<uint64_t>(emu::OP_BALLOT, p ? 1 : 0, 0, 0); }
static inline __attribute__((always_inline)) int __any(int p) { return __ballot(p) != 0; }
static inline __attribute__((always_inline)) int emu_readfirstlane(int v) { return emu::xlane(emu::OP_READFIRST, v, 0, 0); }
static inline __attribute__((always_inline)) int emu_readlane(int v, int l) { return emu::xlane(emu::OP_READLANE, v, l, 0); }
static inline __attribute__((always_inline)) int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound) {
  emu::Fiber *f = emu::cur;
  f->p2 = old; f->p3 = (row_mask & 0xf) | ((bank_mask & 0xf) << 4) | (bound ? 0x100 : 0);
  return emu::xlane(emu::OP_DPP, src, ctrl, 0);
}
// 64-bit DPP (v_mov_b64_dpp; gfx90a+ allows it with row_newbcast only -- the resolver rejects any other control for this form)
static inline __attribute__((always_inline)) double emu_update_dpp(double old, double src, int ctrl, int row_mask, int bank_mask, bool bound) {
  emu::Fiber *f = emu::cur;
  f->old64 = emu::to_bits(old); f->p3 = (row_mask & 0xf) | ((bank_mask & 0xf) << 4) | (bound ? 0x100 : 0);
  return emu::xlane(emu::OP_DPP64, src, ctrl, 0);
}
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef double emu_f64x4 __attribute__((ext_vector_type(4)));
static inline __attribute__((always_inline)) emu_f64x4 emu_mfma_f64_16x16x4f64(double a, double b, emu_f64x4 c, int, int, int) {
  emu::Fiber *f = emu::cur;
  f->op = emu::OP_MFMA_16x16x4_F64;
  f->din[0] = a; f->din[1] = b; f->din[2] = c[0]; f->din[3] = c[1]; f->din[4] = c[2]; f->din[5] = c[3];
  emu::wave_op();
  f = emu::cur;
  emu_f64x4 d = {f->dout[0], f->dout[1], f->dout[2], f->dout[3]};
  return d;
}
static inline __attribute__((always_inline)) emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  emu::Fiber *f = emu::cur;
  f->op = emu::OP_MFMA_16x16x4_F32;
  f->fin[0] = a; f->fin[1] = b; f->fin[2] = c[0]; f->fin[3] = c[1]; f->fin[4] = c[2]; f->fin[5] = c[3];
  emu::wave_op();
  f = emu::cur;
  emu_f32x4 d = {f->fout[0], f->fout[1], f->fout[2], f->fout[3]};
  return d;
}
namespace emu {
static inline __attribute__((always_inline)) float seqsum8(float v) {  // the v_add_f32_dpp chain of csrc/sos_ba.hip:seqsum8
  Fiber *f = cur;
  f->op = OP_SEQSUM8; f->fin[0] = v;
  wave_op();
  return cur->fout[0];
}
}
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
#define __builtin_amdgcn_update_dpp(o, s, c, r, b, bc) emu_update_dpp((o), (s), (c), (r), (b), (bc))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32((a), (b), (c), (x), (y), (z))
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) emu_mfma_f64_16x16x4f64((a), (b), (c), (x), (y), (z))
#define __builtin_amdgcn_wave_barrier() ((void)emu::xlane<int>(emu::OP_WAVE_BARRIER, 0, 0, 0))
#define __builtin_amdgcn_fence(order, scope) std::atomic_thread_fence(std::memory_order_seq_cst)
#define __builtin_amdgcn_s_sleep(n) emu::yield_lane()
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / sqrtf(x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)

// ------------------------------------------------------------------------------------------------ device library
#define warpSize 64
using std::isfinite;
using std::isnan;
using std::isinf;
using std::signbit;
unsigned long long wall_clock64();  // 100 MHz constant clock
static inline unsigned long long clock64() { return wall_clock64(); }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 0
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
template <class T> static inline T emu_atomic_load(const T *p, int order) {
  if constexpr (std::is_floating_point<T>::value) {
    using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    const U u = __atomic_load_n(reinterpret_cast<const U *>(p), order);
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
  } else {
    return __atomic_load_n(p, order);
  }
}
template <class T, class V> static inline void emu_atomic_store(T *p, V val, int order) {
  if constexpr (std::is_floating_point<T>::value) {
    using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    const T t = (T)val;
    U u;
    memcpy(&u, &t, sizeof(T));
    __atomic_store_n(reinterpret_cast<U *>(p), u, order);
  } else {
    __atomic_store_n(p, (T)val, order);
  }
}
#define __hip_atomic_load(p, order, scope) emu_atomic_load((p), (order))
#define __hip_atomic_store(p, v, order, scope) emu_atomic_store((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))

static inline float atomicAdd(float *p, float v) {
  uint32_t *u = reinterpret_cast<uint32_t *>(p), o = __atomic_load_n(u, __ATOMIC_SEQ_CST);
  for (;;) {
    const uint32_t n = emu::from_bits<uint32_t>(emu::to_bits(emu::from_bits<float>(o) + v));
    if (__atomic_compare_exchange_n(u, &o, n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return emu::from_bits<float>(o);
  }
}
static inline double atomicAdd(double *p, double v) {
  uint64_t *u = reinterpret_cast<uint64_t *>(p), o = __atomic_load_n(u, __ATOMIC_SEQ_CST);
  for (;;) {
    const uint64_t n = emu::to_bits(emu::from_bits<double>(o) + v);
    if (__atomic_compare_exchange_n(u, &o, n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return emu::from_bits<double>(o);
  }
}
template <class T, class = typename std::enable_if<std::is_integral<T>::value>::type> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicMax(T *p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
template <class T> static inline T atomicMin(T *p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return o;
}
template <class T> static inline T atomicCAS(T *p, T cmp, T v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}

static inline unsigned __float_as_uint(float f) { return emu::from_bits<unsigned>(emu::to_bits(f)); }
static inline int __float_as_int(float f) { return emu::from_bits<int>(emu::to_bits(f)); }
static inline float __uint_as_float(unsigned u) { return emu::from_bits<float>(u); }
static inline float __int_as_float(int u) { return emu::from_bits<float>((unsigned)u); }
static inline long long __double_as_longlong(double d) { return emu::from_bits<long long>(emu::to_bits(d)); }
static inline double __longlong_as_double(long long d) { return emu::from_bits<double>((uint64_t)d); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline double __fma_rn(double a, double b, double c) { return fma(a, b, c); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
// HIP's global integer min / max overloads
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
