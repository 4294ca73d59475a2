// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE (see include/hip/hip_runtime.h and README.md).
// The run-time half of the lockstep SIMT emulator: streams (one worker thread each, operations in order), memory calls, events,
// and the fiber scheduler that executes a launch: workgroups are admitted up to a residency limit, every thread is a fiber with its
// own stack, the lanes of a wave meet at cross-lane operations and the threads of a workgroup at __syncthreads.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

// void emu_switch(void **save_sp, void *new_sp): callee-saved registers + MXCSR / x87 control word, System V x86-64
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");
extern "C" void emu_switch(void **save_sp, void *new_sp);

// EMU_ASAN build (build_emu.py, EMU_ASAN=1): AddressSanitizer is told about every stack switch, so that out-of-bounds accesses of the
// kernels to "device" memory (hipMalloc = the instrumented malloc), LDS blocks and their own frames are reported instead of going
// unnoticed as they would on the GPU
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile *addr, size_t size);
#endif
#endif

namespace emu {
thread_local Fiber *cur = nullptr;

static long env_long(const char *k, long d) {
  const char *v = getenv(k);
  return v && *v ? strtol(v, nullptr, 0) : d;
}
static const int g_verbose = (int)env_long("EMU_VERBOSE", 0);
static const int g_malloc_fill = (int)env_long("EMU_MALLOC_FILL", 0xFF);
static const int g_shared_fill = (int)env_long("EMU_SHARED_FILL", 0xFF);
static const size_t g_stack_bytes = (size_t)env_long("EMU_STACK_KB", 64) * 1024;
static const long g_resident_threads = env_long("EMU_RESIDENT_THREADS", 65536);
// EMU_SCHED: the order in which the waves of a workgroup (and the lanes of a wave) get to run between meeting points.  0: ascending;
// 1: descending; 2: rotated by a counter.  A kernel that is correct on hardware does not depend on it -- one that misses a barrier
// between an LDS write and a read by another wave does (its tests then differ between EMU_SCHED values).
static const int g_sched = (int)env_long("EMU_SCHED", 0);

[[noreturn]] void die(const char *msg) {
  fprintf(stderr, "[emu] fatal: %s\n", msg);
  fflush(stderr);
  abort();
}

// ------------------------------------------------------------------------------------------------ per-worker machine
struct Machine {
  char *pool = nullptr;
  size_t nslots = 0;
  std::vector<int> free_slots;
  void *sched_sp = nullptr;
  LaunchBase *launch = nullptr;
  size_t high_slot = 0;
  const void *sched_bottom = nullptr;  // (ASan) the worker thread's own stack, learnt at the first switch back
  size_t sched_size = 0;
  ~Machine() {
    if (pool) munmap(pool, nslots * g_stack_bytes);
  }
  void ensure_pool() {
    if (pool) return;
    nslots = (size_t)g_resident_threads;
    void *p = mmap(nullptr, nslots * g_stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) die("mmap of the fiber stack pool failed");
    pool = (char *)p;
    free_slots.reserve(nslots);
    for (size_t i = nslots; i-- > 0;) free_slots.push_back((int)i);
  }
};
static thread_local Machine *tl_machine = nullptr;

struct StaticMap {
  std::unordered_map<const void *, void *> m;
  ~StaticMap() {
    for (auto &kv : m) free(kv.second);
  }
};

void *shared_static_lookup(const void *key, size_t bytes, size_t align) {
  Block *b = cur->blk;
  if (!b->statics) b->statics = new StaticMap();
  auto &m = static_cast<StaticMap *>(b->statics)->m;
  auto it = m.find(key);
  if (it != m.end()) return it->second;
  void *p = aligned_alloc(align < 16 ? 16 : align, (bytes + 63) & ~(size_t)63);
  memset(p, g_shared_fill, bytes);
  m.emplace(key, p);
  b->static_bytes += bytes;  // a workgroup cannot hold more than the 160 KB of LDS of one compute unit (gfx950)
  if (b->static_bytes + b->dyn_bytes > 160 * 1024) die("workgroup needs more than 160 KB of LDS (static + dynamic)");
  return p;
}

static inline void to_scheduler() {
  Fiber *f = cur;
#ifdef EMU_ASAN
  void *fake = nullptr;
  __sanitizer_start_switch_fiber(f->st == DEAD ? nullptr : &fake, tl_machine->sched_bottom, tl_machine->sched_size);
  emu_switch(&f->sp, tl_machine->sched_sp);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
  emu_switch(&f->sp, tl_machine->sched_sp);
#endif
}
__attribute__((noinline)) void wave_op() {
  Fiber *f = cur;
  f->site = __builtin_return_address(0);
  f->st = WAIT_WAVE;
  to_scheduler();
}
void block_sync(int site_id, const char *file, int line) {
  Fiber *f = cur;
  f->bar_id = site_id; f->bar_file = file; f->bar_line = line;
  f->st = WAIT_BLOCK;
  f->blk->waiting_block++;
  to_scheduler();
}
void yield_lane() {
  cur->st = YIELDED;
  to_scheduler();
}

static void fiber_main() {
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &tl_machine->sched_bottom, &tl_machine->sched_size);
#endif
  Fiber *f = cur;
  tl_machine->launch->run_thread();
  f = cur;
  f->st = DEAD;
  f->blk->alive--;
  to_scheduler();
  die("a dead fiber was resumed");
}

static void init_fiber(Fiber *f, char *stack_top) {
  // frame consumed by the tail of emu_switch: [mxcsr|fpcw][r15][r14][r13][r12][rbx][rbp][ret = fiber_main][fake return address]
  uint64_t *s = reinterpret_cast<uint64_t *>(stack_top);
  *--s = 0;                               // fake return address: fiber_main sees rsp % 16 == 8 as after a call
  *--s = (uint64_t)(uintptr_t)&fiber_main;
  for (int i = 0; i < 6; i++) *--s = 0;   // rbp rbx r12 r13 r14 r15
  uint32_t csr[2];
  asm volatile("stmxcsr %0" : "=m"(csr[0]));
  uint16_t cw;
  asm volatile("fnstcw %0" : "=m"(cw));
  csr[1] = cw;
  uint64_t w;
  memcpy(&w, csr, 8);
  *--s = w;
  f->sp = s;
}

// ------------------------------------------------------------------------------------------------ cross-lane resolution
static int dpp_source(int lane, int ctrl, bool *valid) {
  const int row = lane & ~15, l = lane & 15;
  *valid = true;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);  // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = l + (ctrl & 15); *valid = s < 16; return row + (s & 15); }   // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = l - (ctrl & 15); *valid = s >= 0; return row + (s & 15); }   // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12F) return row + ((l - (ctrl & 15)) & 15);                                       // row_ror
  if (ctrl == 0x130) { *valid = lane + 1 < 64; return (lane + 1) & 63; }   // wave_shl:1
  if (ctrl == 0x134) return (lane + 1) & 63;                               // wave_rol:1
  if (ctrl == 0x138) { *valid = lane >= 1; return (lane - 1) & 63; }       // wave_shr:1
  if (ctrl == 0x13C) return (lane - 1) & 63;                               // wave_ror:1
  if (ctrl == 0x140) return row + (15 - l);                                // row_mirror
  if (ctrl == 0x141) return row + ((l & 8) | (7 - (l & 7)));               // row_half_mirror
  if (ctrl == 0x142) { *valid = row >= 16; return row - 1; }               // row_bcast:15 (lane 15 of the previous row)
  if (ctrl == 0x143) { *valid = lane >= 32; return 31; }                   // row_bcast:31
  if (ctrl >= 0x150 && ctrl <= 0x15F) return row + (ctrl & 15);            // row_newbcast:n (gfx90a+): lane n of each row to the whole row
  die("unknown DPP control");
}

static long g_ambiguous = 0, g_inactive_reads = 0;

static void resolve_group(Fiber *w0, int nl, uint64_t mask, int first) {
  Fiber &lead = w0[first];
  const int op = lead.op;
  for (int l = 0; l < nl; l++)
    if ((mask >> l) & 1)
      if (w0[l].op != op) die("lanes of one group wait in different cross-lane operations at the same site");
  auto in_group = [&](int l) { return l >= 0 && l < nl && ((mask >> l) & 1); };
  switch (op) {
    case OP_SHFL: case OP_SHFL_XOR: case OP_SHFL_UP: case OP_SHFL_DOWN: {
      uint64_t outv[64];
      for (int l = 0; l < nl; l++) {
        if (!((mask >> l) & 1)) continue;
        Fiber &f = w0[l];
        const int width = f.p1 > 0 && f.p1 <= 64 ? f.p1 : 64, seg = l & ~(width - 1);
        int src = l;
        if (op == OP_SHFL) src = seg + (f.p0 & (width - 1));
        else if (op == OP_SHFL_XOR) { src = l ^ f.p0; if ((src & ~(width - 1)) != seg) src = l; }
        else if (op == OP_SHFL_UP) { src = l - f.p0; if (src < seg) src = l; }
        else { src = l + f.p0; if (src >= seg + width) src = l; }
        if (in_group(src)) outv[l] = w0[src].in64;
        else { outv[l] = 0; g_inactive_reads++; }
      }
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].out64 = outv[l];
      break;
    }
    case OP_BALLOT: {
      uint64_t b = 0;
      for (int l = 0; l < nl; l++) if (((mask >> l) & 1) && w0[l].in64) b |= 1ull << l;
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].out64 = b;
      break;
    }
    case OP_READFIRST: {
      const uint64_t v = lead.in64;
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].out64 = v;
      break;
    }
    case OP_READLANE: {
      uint64_t outv[64];
      for (int l = 0; l < nl; l++) {
        if (!((mask >> l) & 1)) continue;
        const int src = w0[l].p0 & 63;
        if (in_group(src)) outv[l] = w0[src].in64;
        else { outv[l] = 0; g_inactive_reads++; }
      }
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].out64 = outv[l];
      break;
    }
    case OP_DPP: {
      uint64_t outv[64];
      for (int l = 0; l < nl; l++) {
        if (!((mask >> l) & 1)) continue;
        Fiber &f = w0[l];
        const int ctrl = f.p0, rmask = f.p3 & 0xf, bmask = (f.p3 >> 4) & 0xf;
        const bool bound = (f.p3 & 0x100) != 0;
        const uint64_t oldv = (uint32_t)f.p2;
        if (!((rmask >> (l >> 4)) & 1) || !((bmask >> ((l >> 2) & 3)) & 1)) { outv[l] = oldv; continue; }
        bool valid;
        const int src = dpp_source(l, ctrl, &valid);
        if (valid && in_group(src)) outv[l] = w0[src].in64;
        else outv[l] = bound ? 0 : oldv;
      }
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].out64 = outv[l];
      break;
    }
    case OP_DPP64: {  // v_mov_b64_dpp: the DP-ALU DPP form exists with row_newbcast only
      uint64_t outv[64];
      for (int l = 0; l < nl; l++) {
        if (!((mask >> l) & 1)) continue;
        Fiber &f = w0[l];
        const int ctrl = f.p0, rmask = f.p3 & 0xf, bmask = (f.p3 >> 4) & 0xf;
        const bool bound = (f.p3 & 0x100) != 0;
        if (ctrl < 0x150 || ctrl > 0x15F) die("64-bit DPP with a control other than row_newbcast");
        if (!((rmask >> (l >> 4)) & 1) || !((bmask >> ((l >> 2) & 3)) & 1)) { outv[l] = f.old64; continue; }
        bool valid;
        const int src = dpp_source(l, ctrl, &valid);
        if (valid && in_group(src)) outv[l] = w0[src].in64;
        else outv[l] = bound ? 0 : f.old64;
      }
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].out64 = outv[l];
      break;
    }
    case OP_SEQSUM8: {
      float outv[64];
      for (int l = 0; l < nl; l++) {
        if (!((mask >> l) & 1)) continue;
        const int row = l & ~15, i = l & 15;
        auto src = [&](int k) -> float { return (i >= k && in_group(row + i - k)) ? w0[row + i - k].fin[0] : 0.0f; };
        volatile float s = src(7) + 0.0f;
        for (int k = 6; k >= 1; k--) s = src(k) + s;
        s = s + w0[l].fin[0];
        outv[l] = s;
      }
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].fout[0] = outv[l];
      break;
    }
    case OP_MFMA_16x16x4_F32: {
      if (nl != 64 || mask != ~0ull) die("MFMA executed by a partial wave");
      float D[16][16];
      for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
          float acc = w0[16 * (i / 4) + j].fin[2 + (i & 3)];
          for (int k = 0; k < 4; k++) acc = fmaf(w0[16 * k + i].fin[0], w0[16 * k + j].fin[1], acc);
          D[i][j] = acc;
        }
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) w0[l].fout[r] = D[4 * (l / 16) + r][l & 15];
      break;
    }
    case OP_MFMA_16x16x4_F64: {
      // v_mfma_f64_16x16x4_f64: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k (as the f32 16x16x4 form), but C / D with
      // row = (lane >> 4) + 4 * reg, col = lane & 15 -- NOT the f32 row formula (cdna_hip_programming.md, fragment layout)
      if (nl != 64 || mask != ~0ull) die("MFMA executed by a partial wave");
      double D[16][16];
      for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
          double acc = w0[16 * (i & 3) + j].din[2 + (i >> 2)];
          for (int k = 0; k < 4; k++) acc = fma(w0[16 * k + i].din[0], w0[16 * k + j].din[1], acc);
          D[i][j] = acc;
        }
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) w0[l].dout[r] = D[(l >> 4) + 4 * r][l & 15];
      break;
    }
    case OP_WAVE_BARRIER: break;
    default: die("unknown cross-lane operation");
  }
  for (int l = 0; l < nl; l++) if ((mask >> l) & 1) w0[l].st = RUNNABLE;
}

// ------------------------------------------------------------------------------------------------ launch execution
static inline void run_fiber(Machine *M, Fiber *f) {
  cur = f;
#ifdef EMU_ASAN
  void *fake = nullptr;
  const size_t slot = (size_t)(((char *)f->sp - M->pool) / g_stack_bytes);
  __sanitizer_start_switch_fiber(&fake, M->pool + slot * g_stack_bytes, g_stack_bytes);
  emu_switch(&M->sched_sp, f->sp);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
  emu_switch(&M->sched_sp, f->sp);
#endif
}

// one scheduling pass over a block; returns true if anything moved
static bool pass_block(Machine *M, Block *b, bool *only_yield) {
  bool progressed = false;
  static thread_local unsigned rot = 0;
  rot += 7;
  for (int wq = 0; wq < b->nwaves; wq++) {
    const int w = g_sched == 0 ? wq : g_sched == 1 ? b->nwaves - 1 - wq : (int)((wq + rot) % (unsigned)b->nwaves);
    Fiber *w0 = b->fibers + (size_t)w * 64;
    const int nl = std::min(64, b->nthreads - w * 64);
    for (int l = 0; l < nl; l++) if (w0[l].st == YIELDED) { w0[l].st = RUNNABLE; }
    for (;;) {
      bool ran = false, yielded_only = true;
      for (int lq = 0; lq < nl; lq++) {
        const int l = g_sched == 0 ? lq : g_sched == 1 ? nl - 1 - lq : (int)((lq + rot) % (unsigned)nl);
        if (w0[l].st != RUNNABLE) continue;
        run_fiber(M, &w0[l]);
        ran = true;
        if (w0[l].st != YIELDED) yielded_only = false;
      }
      if (ran) { progressed = true; if (!yielded_only) *only_yield = false; }
      // nobody runnable now: resolve the waiting cross-lane group at the lowest site
      const void *site = nullptr;
      int first = -1;
      for (int l = 0; l < nl; l++) {
        if (w0[l].st != WAIT_WAVE) continue;
        if (!site || (uintptr_t)w0[l].site < (uintptr_t)site) { site = w0[l].site; first = l; }
      }
      if (first < 0) break;
      uint64_t mask = 0;
      bool other = false;
      for (int l = 0; l < nl; l++) {
        if (w0[l].st != WAIT_WAVE) continue;
        if (w0[l].site == site) mask |= 1ull << l;
        else other = true;
      }
      if (other) {
        g_ambiguous++;
        if (g_verbose > 1) fprintf(stderr, "[emu] %s: lanes of one wave wait at different cross-lane sites (%p first)\n", M->launch->name, site);
      }
      // the first lane of the group in lane order leads (readfirstlane)
      for (int l = 0; l < nl; l++) if ((mask >> l) & 1) { first = l; break; }
      resolve_group(w0, nl, mask, first);
      progressed = true;
      *only_yield = false;
    }
  }
  if (b->alive > 0 && b->waiting_block == b->alive) {
    // the barrier releases.  Wave-granular check: all waiting lanes of a wave are in the SAME __syncthreads statement (different waves
    // may legitimately stand in different ones -- s_barrier is anonymous --, the lanes of one wave may not: see hip_runtime.h)
    for (int w = 0; w < b->nwaves; w++) {
      const Fiber *first = nullptr;
      for (int l = 0; l < 64 && w * 64 + l < b->nthreads; l++) {
        const Fiber *f = &b->fibers[w * 64 + l];
        if (f->st != WAIT_BLOCK) continue;
        if (!first) first = f;
        else if (f->bar_id != first->bar_id || f->bar_file != first->bar_file) {
          fprintf(stderr, "[emu] %s block (%u,%u,%u) wave %d: lane %d waits in the __syncthreads of %s:%d, lane %d in that of %s:%d\n", M->launch->name, b->idx.x,
                  b->idx.y, b->idx.z, w, first->lane, first->bar_file, first->bar_line, f->lane, f->bar_file, f->bar_line);
          die("divergent __syncthreads: lanes of one wave wait in different barrier statements (on hardware the wave executes an s_barrier for each)");
        }
      }
    }
    int o = 0, c = 0;
    for (int t = 0; t < b->nthreads; t++)
      if (b->fibers[t].st == WAIT_BLOCK) { o |= b->fibers[t].pred != 0; c += b->fibers[t].pred != 0; }
    b->sync_res_or = o;
    b->sync_res_count = c;
    for (int t = 0; t < b->nthreads; t++)
      if (b->fibers[t].st == WAIT_BLOCK) { b->fibers[t].st = RUNNABLE; b->fibers[t].pred = 0; }
    b->waiting_block = 0;
    progressed = true;
    *only_yield = false;
  }
  return progressed;
}

static void run_launch(Machine *M, LaunchBase *L) {
  M->ensure_pool();
  M->launch = L;
  const dim3 g = L->grid, bd = L->block;
  const long nthreads = (long)bd.x * bd.y * bd.z;
  const long nblocks = (long)g.x * g.y * g.z;
  if (nthreads <= 0 || nthreads > 1024) die("block size out of range");
  if (L->shmem > 160 * 1024) die("launch asks for more than 160 KB of dynamic LDS");
  if (nblocks <= 0) { M->launch = nullptr; return; }
  const long max_resident = std::max(1L, (long)M->nslots / nthreads);
  std::vector<Block *> resident;
  long next = 0;
  const auto t0 = std::chrono::steady_clock::now();
  long idle_passes = 0;
  while (next < nblocks || !resident.empty()) {
    while (next < nblocks && (long)resident.size() < max_resident) {
      Block *b = new Block();
      b->idx = dim3((unsigned)(next % g.x), (unsigned)((next / g.x) % g.y), (unsigned)(next / ((long)g.x * g.y)));
      b->bdim = bd; b->gdim = g;
      b->nthreads = (int)nthreads; b->nwaves = (int)((nthreads + 63) / 64);
      b->alive = (int)nthreads; b->waiting_block = 0;
      b->fibers = new Fiber[nthreads];
      b->dyn_bytes = L->shmem;
      b->dyn_shared = nullptr;
      if (L->shmem) { b->dyn_shared = aligned_alloc(64, (L->shmem + 63) & ~(size_t)63); memset(b->dyn_shared, g_shared_fill, L->shmem); }
      b->statics = nullptr;
      b->sync_res_or = b->sync_res_count = 0;
      for (long t = 0; t < nthreads; t++) {
        Fiber *f = &b->fibers[t];
        memset(f, 0, sizeof(Fiber));
        f->tid = uint3{(unsigned)(t % bd.x), (unsigned)((t / bd.x) % bd.y), (unsigned)(t / ((long)bd.x * bd.y))};
        f->blk = b; f->lin = (int)t; f->lane = (int)(t & 63); f->wave = (int)(t >> 6); f->st = RUNNABLE;
        const int slot = M->free_slots.back();
        M->free_slots.pop_back();
        if ((size_t)slot > M->high_slot) M->high_slot = slot;
#ifdef EMU_ASAN
        // a fiber that ended left the redzones of its last frames poisoned in the shadow of this slot
        __asan_unpoison_memory_region(M->pool + (size_t)slot * g_stack_bytes, g_stack_bytes);
#endif
        init_fiber(f, M->pool + ((size_t)slot + 1) * g_stack_bytes);  // the slot is recovered from f->sp when the block retires
      }
      resident.push_back(b);
      next++;
    }
    bool any = false, only_yield = true;
    for (size_t i = 0; i < resident.size();) {
      Block *b = resident[i];
      any |= pass_block(M, b, &only_yield);
      if (b->alive == 0) {
        for (int t = 0; t < b->nthreads; t++) {
          const size_t slot = (size_t)(((char *)b->fibers[t].sp - M->pool) / g_stack_bytes);
          M->free_slots.push_back((int)slot);
        }
        delete[] b->fibers;
        free(b->dyn_shared);
        delete static_cast<StaticMap *>(b->statics);
        delete b;
        resident[i] = resident.back();
        resident.pop_back();
        any = true;
      } else {
        i++;
      }
    }
    if (!any) {
      fprintf(stderr, "[emu] deadlock in %s: %zu resident blocks, no thread can move\n", L->name, resident.size());
      for (Block *b : resident) {
        int st[5] = {0, 0, 0, 0, 0};
        for (int t = 0; t < b->nthreads; t++) st[b->fibers[t].st]++;
        fprintf(stderr, "  block (%u,%u,%u): runnable %d wave-wait %d block-wait %d yielded %d dead %d\n", b->idx.x, b->idx.y, b->idx.z, st[0], st[1], st[2], st[3], st[4]);
      }
      die("deadlock");
    }
    if (only_yield) {
      // every movable thread is polling (s_sleep): let the host / other streams run
      if (++idle_passes > 16) std::this_thread::sleep_for(std::chrono::microseconds(20));
      else sched_yield();
    } else {
      idle_passes = 0;
    }
  }
  if (M->high_slot > 2048) {  // give the touched stack pages of a wide launch back
    madvise(M->pool + 2048 * g_stack_bytes, (M->high_slot + 1 - 2048) * g_stack_bytes, MADV_DONTNEED);
    M->high_slot = 0;
  }
  if (g_verbose) {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "[emu] %-28s grid %ld x %ld lds %zu  %.2f ms\n", L->name, nblocks, nthreads, L->shmem, ms);
  }
  M->launch = nullptr;
}

// ------------------------------------------------------------------------------------------------ streams
struct Stream {
  std::thread th;
  std::mutex m;
  std::condition_variable cv, cv_idle;
  std::deque<std::function<void()>> q;
  bool busy = false, stop = false;
  Machine mach;
  Stream() { th = std::thread([this] { loop(); }); }
  ~Stream() {
    {
      std::unique_lock<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void loop() {
    tl_machine = &mach;
    for (;;) {
      std::function<void()> fn;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) return;
        fn = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      fn();
      {
        std::unique_lock<std::mutex> lk(m);
        busy = false;
        if (q.empty()) cv_idle.notify_all();
      }
    }
  }
  void push(std::function<void()> fn) {
    {
      std::unique_lock<std::mutex> lk(m);
      q.push_back(std::move(fn));
    }
    cv.notify_one();
  }
  void sync() {
    std::unique_lock<std::mutex> lk(m);
    cv_idle.wait(lk, [this] { return q.empty() && !busy; });
  }
  bool idle() {
    std::unique_lock<std::mutex> lk(m);
    return q.empty() && !busy;
  }
};
struct Event {
  std::mutex m;
  std::condition_variable cv;
  long pending = 0;
  std::chrono::steady_clock::time_point t;
};

static std::mutex g_streams_m;
static std::vector<Stream *> g_streams;
static Stream *g_default = nullptr;
// nullptr = the default stream; a handle that is not (or no longer) registered resolves to nullptr -> hipErrorInvalidValue, as the
// real run time answers a stale handle (objects finalised in arbitrary order at interpreter exit) instead of touching freed memory
static Stream *resolve(hipStream_t st) {
  std::unique_lock<std::mutex> lk(g_streams_m);
  if (st) {
    for (Stream *s : g_streams) if (s == st) return s;
    return nullptr;
  }
  if (!g_default) { g_default = new Stream(); g_streams.push_back(g_default); }
  return g_default;
}
static void sync_all() {
  std::vector<Stream *> v;
  {
    std::unique_lock<std::mutex> lk(g_streams_m);
    v = g_streams;
  }
  for (Stream *s : v) s->sync();
}

void enqueue_launch(hipStream_t st, LaunchBase *l) {
  Stream *s = resolve(st);
  if (!s) { delete l; return; }
  s->push([s, l] {
    run_launch(&s->mach, l);
    delete l;
  });
}
}  // namespace emu

using emu::Stream;
using emu::Event;

unsigned long long wall_clock64() {
  return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "lockstep CPU emulation (tests/emu) -- not a GPU");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu-gfx950");
  p->totalGlobalMem = (size_t)16 << 30;
  p->sharedMemPerBlock = 160 * 1024;
  p->multiProcessorCount = (int)emu::env_long("EMU_CUS", 256);
  p->wavefrontWidth = 64;
  p->maxThreadsPerBlock = 1024;
  p->clockRate = 2400000;
  p->major = 9; p->minor = 5;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) {
  if (a == hipDeviceAttributeMultiprocessorCount) { *v = (int)emu::env_long("EMU_CUS", 256); return hipSuccess; }
  if (a == hipDeviceAttributeMaxSharedMemoryPerBlock) { *v = 160 * 1024; return hipSuccess; }
  return hipErrorInvalidValue;
}
hipError_t hipDeviceSynchronize() { emu::sync_all(); return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }
hipError_t emu_hipMalloc(void **p, size_t n) {
  void *q = aligned_alloc(256, (n + 255) & ~(size_t)255);
  if (!q && n) return hipErrorOutOfMemory;
  if (q) memset(q, emu::g_malloc_fill, n);
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  emu::sync_all();
  free(p);
  return hipSuccess;
}
hipError_t emu_hipHostMalloc(void **p, size_t n, unsigned) {
  void *q = aligned_alloc(256, (n + 255) & ~(size_t)255);
  if (!q && n) return hipErrorOutOfMemory;
  if (q) memset(q, 0, n);
  *p = q;
  return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  emu::sync_all();
  free(p);
  return hipSuccess;
}
hipError_t emu_hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) {
  emu::sync_all();  // the blocking form waits for everything that may produce the source
  memmove(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t st) {
  Stream *s = emu::resolve(st);
  if (!s) return hipErrorInvalidValue;
  if (k == hipMemcpyHostToDevice) {  // pageable sources are staged at the call; a correct caller cannot tell the difference
    std::vector<char> *stage = new std::vector<char>((const char *)src, (const char *)src + n);
    s->push([dst, stage] {
      memcpy(dst, stage->data(), stage->size());
      delete stage;
    });
  } else {
    s->push([dst, src, n] { memmove(dst, src, n); });
  }
  return hipSuccess;
}
// a host function in stream order (what the real run time offers under the same name): tests/emu/fake_rccl.cpp enqueues its collectives
// with it, so that they run where ncclAllReduce would -- behind the kernels in front of them on the stream
hipError_t hipLaunchHostFunc(hipStream_t st, void (*fn)(void *), void *user) {
  Stream *s = emu::resolve(st);
  if (!s || !fn) return hipErrorInvalidValue;
  s->push([fn, user] { fn(user); });
  return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) {
  emu::sync_all();
  memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t st) {
  Stream *s = emu::resolve(st);
  if (!s) return hipErrorInvalidValue;
  s->push([p, v, n] { memset(p, v, n); });
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned) {
  Stream *s = new Stream();
  {
    std::unique_lock<std::mutex> lk(emu::g_streams_m);
    emu::g_streams.push_back(s);
  }
  *st = s;
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *st) { return hipStreamCreateWithFlags(st, 0); }
hipError_t hipStreamDestroy(hipStream_t st) {
  if (!st) return hipSuccess;
  if (!emu::resolve(st)) return hipErrorInvalidValue;
  st->sync();
  {
    std::unique_lock<std::mutex> lk(emu::g_streams_m);
    for (size_t i = 0; i < emu::g_streams.size(); i++)
      if (emu::g_streams[i] == st) { emu::g_streams.erase(emu::g_streams.begin() + i); break; }
  }
  delete st;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t st) {
  Stream *s = emu::resolve(st);
  if (!s) return hipErrorInvalidValue;
  s->sync();
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t st) {
  Stream *s = emu::resolve(st);
  if (!s) return hipErrorInvalidValue;
  return s->idle() ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new Event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new Event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) {
  if (e) { hipEventSynchronize(e); delete e; }
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  {
    std::unique_lock<std::mutex> lk(e->m);
    e->pending++;
  }
  Stream *s = emu::resolve(st);
  if (!s) { std::unique_lock<std::mutex> lk(e->m); e->pending--; return hipErrorInvalidValue; }
  s->push([e] {
    std::unique_lock<std::mutex> lk(e->m);
    e->t = std::chrono::steady_clock::now();
    e->pending--;
    e->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> lk(e->m);
  e->cv.wait(lk, [e] { return e->pending == 0; });
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  hipEventSynchronize(a);
  hipEventSynchronize(b);
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
// statistics of the session (tests/emu/conftest prints them): cross-lane reads of lanes outside the executing group, waves whose
// lanes waited at different sites at once
void emu_stats(long *ambiguous, long *inactive_reads) {
  *ambiguous = emu::g_ambiguous;
  *inactive_reads = emu::g_inactive_reads;
}
}
