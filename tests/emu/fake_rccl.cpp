// tests/emu/fake_rccl.cpp -- TEST INFRASTRUCTURE: the six RCCL entry points csrc/sos_comm.hip binds (dlopen + dlsym), for ranks that are
// PROCESSES of the emulated device library (one emulator per process, as one GPU per process in production).  The collectives are what
// ncclAllReduce / ncclAllGather are on a stream: enqueued behind the kernels in front of them (hipLaunchHostFunc of the emulator's run
// time), executed when the stream gets there, every rank blocking in them until all ranks have arrived.  The ranks meet in a POSIX
// shared-memory segment named by the unique id: a sense-reversing barrier on two atomics and one slot of FAKE_SLOT_BYTES per rank.
// The sum is formed in RANK ORDER by every rank for itself (same operands, same order: bit-identical results on all ranks, as the
// stitch that follows needs).  Nothing of the product links or names this file; sos_rccl_load(path) is given its path by the tests.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

#include <dlfcn.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#define FAKE_SLOT_BYTES ((size_t)8 << 20)
#define FAKE_MAX_RANKS 8

namespace {
struct Ctrl {
  std::atomic<int> arrived, sense, joined;
  int nranks;
};
struct Comm {
  Ctrl *ctl;
  char *slots;
  int nranks, rank, local_sense;
  size_t bytes;
  char name[64];
};
struct Op {
  Comm *c;
  const void *send;
  void *recv;
  size_t count;
  int dtype, kind, op;  // kind 0 all-reduce, 1 all-gather
};

void barrier(Comm *c) {
  c->local_sense ^= 1;
  if (c->ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
    c->ctl->arrived.store(0, std::memory_order_relaxed);
    c->ctl->sense.store(c->local_sense, std::memory_order_release);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->sense.load(std::memory_order_acquire) != c->local_sense) {
      std::this_thread::sleep_for(std::chrono::microseconds(20));
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
        fprintf(stderr, "[fake_rccl] rank %d: the other ranks did not arrive at a collective within 120 s\n", c->rank);
        abort();
      }
    }
  }
}
size_t elem(int dtype) { return dtype == ncclFloat64 || dtype == ncclInt64 || dtype == ncclUint64 ? 8 : dtype == ncclInt8 || dtype == ncclUint8 ? 1 : 4; }

template <class T> void reduce(Comm *c, T *out, size_t n, int op) {
  for (size_t i = 0; i < n; i++) {
    T a = reinterpret_cast<const T *>(c->slots)[i];
    for (int r = 1; r < c->nranks; r++) {
      const T b = reinterpret_cast<const T *>(c->slots + (size_t)r * FAKE_SLOT_BYTES)[i];
      a = op == ncclMax ? (b > a ? b : a) : op == ncclMin ? (b < a ? b : a) : a + b;
    }
    out[i] = a;
  }
}
void run(void *p) {
  Op *o = static_cast<Op *>(p);
  Comm *c = o->c;
  const size_t bytes = o->count * elem(o->dtype);
  if (bytes > FAKE_SLOT_BYTES) { fprintf(stderr, "[fake_rccl] message of %zu bytes exceeds the slot\n", bytes); abort(); }
  memcpy(c->slots + (size_t)c->rank * FAKE_SLOT_BYTES, o->send, bytes);
  barrier(c);  // every rank's contribution is in its slot
  if (o->kind == 1) {
    for (int r = 0; r < c->nranks; r++) memcpy(static_cast<char *>(o->recv) + (size_t)r * bytes, c->slots + (size_t)r * FAKE_SLOT_BYTES, bytes);
  } else if (o->dtype == ncclFloat32) {
    reduce(c, static_cast<float *>(o->recv), o->count, o->op);
  } else if (o->dtype == ncclFloat64) {
    reduce(c, static_cast<double *>(o->recv), o->count, o->op);
  } else if (o->dtype == ncclInt32) {
    reduce(c, static_cast<int *>(o->recv), o->count, o->op);
  } else {
    fprintf(stderr, "[fake_rccl] data type %d not emulated\n", o->dtype);
    abort();
  }
  barrier(c);  // everybody has read the slots: the next collective may overwrite them
  delete o;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/sos_fake_rccl_%d_%lld", (int)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm *c = new Comm();
  c->nranks = nranks; c->rank = rank; c->local_sense = 0;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->bytes = 4096 + (size_t)nranks * FAKE_SLOT_BYTES;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return ncclSystemError; }
  if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return ncclSystemError; }  // (a fresh segment is zero: the counters start at 0)
  void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return ncclSystemError; }
  c->ctl = static_cast<Ctrl *>(m);
  c->slots = static_cast<char *>(m) + 4096;
  c->ctl->joined.fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->ctl->joined.load() < nranks) {  // as ncclCommInitRank: returns when all ranks have joined
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { munmap(m, c->bytes); delete c; return ncclSystemError; }
  }
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t h) {
  Comm *c = reinterpret_cast<Comm *>(h);
  if (!c) return ncclSuccess;
  if (c->ctl->joined.fetch_sub(1) == 1) shm_unlink(c->name);  // the last one out removes the name
  munmap(c->ctl, c->bytes);
  delete c;
  return ncclSuccess;
}
// the EMULATOR's hipLaunchHostFunc, looked up in the emulated device library by its path (SOS_FAKE_RCCL_EMU_LIB, set by the tests): a
// process may also hold the real HIP run time (torch), whose function of the same name knows nothing of the emulated streams
static ncclResult_t enqueue(Op *o, hipStream_t st) {
  typedef hipError_t (*launch_t)(hipStream_t, void (*)(void *), void *);
  static launch_t launch = nullptr;
  if (!launch) {
    const char *path = getenv("SOS_FAKE_RCCL_EMU_LIB");
    void *h = path ? dlopen(path, RTLD_NOW | RTLD_NOLOAD) : nullptr;
    launch = h ? reinterpret_cast<launch_t>(dlsym(h, "hipLaunchHostFunc")) : nullptr;
    if (!launch) { fprintf(stderr, "[fake_rccl] SOS_FAKE_RCCL_EMU_LIB does not name the loaded emulated device library\n"); delete o; return ncclInvalidUsage; }
  }
  return launch(st, run, o) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t h, hipStream_t st) {
  return enqueue(new Op{reinterpret_cast<Comm *>(h), send, recv, count, (int)dt, 0, (int)op}, st);
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t h, hipStream_t st) {
  return enqueue(new Op{reinterpret_cast<Comm *>(h), send, recv, sendcount, (int)dt, 1, 0}, st);
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ncclSuccess" : "emulated RCCL error"; }
}
