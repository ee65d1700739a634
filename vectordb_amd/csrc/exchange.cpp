// eps_exchange: the ONE exchange step of the sharded path, owned by the library.  One process per GPU; every rank holds the
// top-k lists of the whole batch over ITS shard of the table (global ids), packed as [ids int64[nq][k] | dist f32[nq][k]]; the lists meet
// in one ncclAllGather over the RCCL communicator this file creates (xGMI inside a node) and every rank merges the `world` sorted lists of
// each query by (dist, id) - merge_shards_kernel, the kernel eps_merge_topk_packed launches.  No other collective exists on the path.
//
// RCCL is resolved at run time (dlopen librccl.so.1, then librccl.so): a single-GPU user of the library does not need it, and in a process
// that has already loaded a copy (PyTorch ships one) that copy is the one used - two RCCLs in one process would each want the devices' IPC state.
// The caller's only job is the bootstrap: rank 0 obtains 128 bytes from eps_exchange_unique_id and hands them to every rank by whatever it has
// (bench.py: a gloo broadcast; a DBMS: its own RPC).  Nothing here falls back to anything: a failure is a status code and a message.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "epsilla_gfx950.h"
#include "index.hpp"
#include "kernels.hpp"

namespace {

struct Rccl {
  void* lib = nullptr;
  std::string path, err;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  int version = 0;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) {
        r.path = name;
        break;
      }
    }
    if (!r.lib) {
      const char* e = dlerror();
      r.err = std::string("RCCL not found (librccl.so.1 / librccl.so): ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(r.lib, n);
      if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (r.err.empty() && r.GetVersion) (void)r.GetVersion(&r.version);
    Dl_info info;
    if (r.GetUniqueId && dladdr(reinterpret_cast<void*>(r.GetUniqueId), &info) && info.dli_fname) r.path = info.dli_fname;   // (the copy that really answers)
  });
  return r;
}

}  // namespace

// ---- the b = 1 form ("for b=1 prefer direct P2P stores into a peer-mapped buffer + flag").  Every rank owns a MAILBOX on its device:
//   u64 seq[16]                          seq[r] = number of the last call whose lists rank r has delivered here
//   char slot[2][world][MB_SLOT]         rank r's packed lists [ids int64[nk] | dist f32[nk]] of call c live in slot[c & 1][r]
// mapped by every peer (hipIpc).  A call = one push launch (block p stores this rank's lists into peer p's slot, fences at system scope, then
// raises seq[rank] there) + one wait launch (spins until all `world` entries of the OWN seq show the call's number) + merge_shards_kernel over the own
// slots.  No collective, no host round trip.  Two parities suffice: a rank leaves the wait of call c only after every peer has pushed c, i.e.
// finished (in stream order) merging c - 1, so the slots of parity (c - 1) are free when anybody pushes c + 1.
constexpr size_t MB_SLOT = 16384;      // bytes of one rank's lists of one call: nq * k * 12 <= MB_SLOT (k = 10: 136 queries)
constexpr size_t MB_HEAD = 128;        // u64 seq[16]
constexpr int MB_MAX_WORLD = 16;
struct MboxPeers {
  char* p[MB_MAX_WORLD];
};
__global__ void mbox_push_kernel(MboxPeers peers, int rank, int world, unsigned long long call, const int64_t* ids, const float* dist, int64_t nk) {
  char* base = peers.p[blockIdx.x];
  char* dst = base + MB_HEAD + ((call & 1ull) * (size_t)world + (size_t)rank) * MB_SLOT;
  int64_t* di = reinterpret_cast<int64_t*>(dst);
  float* dd = reinterpret_cast<float*>(dst + nk * 8);
  for (int64_t i = threadIdx.x; i < nk; i += blockDim.x) {
    di[i] = ids[i];
    dd[i] = dist[i];
  }
  __threadfence_system();   // every thread's stores before ...
  __syncthreads();
  if (threadIdx.x == 0)     // ... the flag (release at system scope: a peer that sees it sees the lists)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base) + rank, call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one wavefront: lane r waits for rank r's flag.  Bounded (a peer that died must not hang the stream for ever): *timeout is set and the merge runs on what there is.
__global__ void mbox_wait_kernel(const char* own, int world, unsigned long long call, unsigned* timeout, unsigned long long max_ticks) {
  const int r = threadIdx.x;
  if (r >= world) return;
  const unsigned long long* seq = reinterpret_cast<const unsigned long long*>(own) + r;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < call) {
    if (wall_clock64() - t0 > max_ticks) {
      atomicOr(timeout, 1u << r);
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  __threadfence_system();
}

struct eps_exchange {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  // b = 1 form
  char* mbox = nullptr;            // own mailbox (device memory, exported)
  bool mbox_fine = false;
  char* peer[MB_MAX_WORLD] = {};   // every rank's mailbox as this process sees it (peer[rank] = mbox)
  bool connected = false;
  unsigned long long direct_calls = 0;
  unsigned* timeout_dev = nullptr; // [1] bit r: rank r's lists did not arrive in time
  eps::DevBuf send, gathered;
  static constexpr int RING = 64;   // event triples of the last calls (read back after a timed region: no sync inside it)
  hipEvent_t ev[RING][3] = {};
  int64_t calls = 0;
  std::string err;
  int32_t fail(int32_t code, const std::string& m) {
    err = m;
    return code;
  }
};

extern "C" {

int32_t eps_exchange_unique_id(void* id128) {
  if (!id128) return EPS_USER_ERROR;
  Rccl& r = rccl();
  if (!r.err.empty()) return EPS_INFRA_UNEXPECTED_ERROR;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return EPS_INFRA_UNEXPECTED_ERROR;
  static_assert(sizeof(id) == EPS_EXCHANGE_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof id);
  return EPS_OK;
}

int32_t eps_exchange_create(int32_t rank, int32_t world, const void* id128, int32_t device, eps_exchange** out) {
  if (!out) return EPS_USER_ERROR;
  *out = nullptr;
  eps_exchange* x = new eps_exchange();
  *out = x;   // (handed out even on failure: the message lives in it)
  x->rank = rank;
  x->world = world;
  x->device = device;
  if (!id128 || world < 1 || world > 16 || rank < 0 || rank >= world) return x->fail(EPS_USER_ERROR, "eps_exchange_create: rank / world (1..16) / unique id");
  Rccl& r = rccl();
  if (!r.err.empty()) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, r.err);
  if (hipSetDevice(device) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_create: hipSetDevice");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  const ncclResult_t rc = r.CommInitRank(&x->comm, world, id, rank);
  if (rc != ncclSuccess) {
    x->comm = nullptr;
    return x->fail(EPS_INFRA_UNEXPECTED_ERROR, std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
  }
  for (auto& t : x->ev)
    for (auto& e : t)
      if (hipEventCreate(&e) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_create: hipEventCreate");
  return EPS_OK;
}

int32_t eps_exchange_allgather_merge(eps_exchange* x, const int64_t* ids, const float* dist, int64_t nq, int32_t k, int64_t* out_ids, float* out_dist,
                                     void* hip_stream) {
  if (!x) return EPS_USER_ERROR;
  if (!x->comm) return x->fail(EPS_USER_ERROR, "eps_exchange: no communicator (create failed)");
  if (!ids || !dist || !out_ids || !out_dist || nq < 0 || k <= 0) return x->fail(EPS_USER_ERROR, "eps_exchange_allgather_merge: null pointer / nq / k");
  if (nq == 0) return EPS_OK;
  if (!eps::is_device_ptr(ids) || !eps::is_device_ptr(dist) || !eps::is_device_ptr(out_ids) || !eps::is_device_ptr(out_dist))
    return x->fail(EPS_USER_ERROR, "eps_exchange_allgather_merge: lists and results live on the rank's device");
  if (hipSetDevice(x->device) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  Rccl& r = rccl();
  const size_t nk = (size_t)nq * (size_t)k;
  const size_t stride = (nk * 12 + 7) / 8 * 8;   // [ids int64[nk] | dist f32[nk]], 8-byte aligned per rank
  if (!x->gathered.reserve(stride * (size_t)x->world)) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange: out of device memory");
  // the rank's own lists packed into ITS slot of the gathered buffer: the all-gather runs in place (no send buffer, no extra copy of the largest part)
  char* mine = static_cast<char*>(x->gathered.p) + stride * (size_t)x->rank;
  const bool packed = reinterpret_cast<const char*>(dist) == reinterpret_cast<const char*>(ids) + nk * 8 && nk * 12 == stride;   // the caller's buffer already has the layout (and its length)
  const void* src = mine;
  if (packed && x->world > 1) {
    src = ids;   // out of place straight from the caller's packed buffer
  } else {
    if (hipMemcpyAsync(mine, ids, nk * 8, hipMemcpyDeviceToDevice, s) != hipSuccess || hipMemcpyAsync(mine + nk * 8, dist, nk * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange: pack copy");
  }
  hipEvent_t* ev = x->ev[x->calls % eps_exchange::RING];
  (void)hipEventRecord(ev[0], s);
  const ncclResult_t rc = r.AllGather(src, x->gathered.p, stride, ncclUint8, x->comm, s);
  if (rc != ncclSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, std::string("ncclAllGather: ") + r.GetErrorString(rc));
  (void)hipEventRecord(ev[1], s);
  const char* base = static_cast<const char*>(x->gathered.p);
  eps::launch_merge_shards(reinterpret_cast<const float*>(base + nk * 8), reinterpret_cast<const int64_t*>(base), x->world, nq, k, out_dist, out_ids, s, (int64_t)stride);
  (void)hipEventRecord(ev[2], s);
  x->calls += 1;
  return hipGetLastError() == hipSuccess ? EPS_OK : x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange: merge launch");
}

static int32_t mbox_alloc(eps_exchange* x) {
  if (x->mbox) return EPS_OK;
  if (hipSetDevice(x->device) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
  const size_t bytes = MB_HEAD + 2 * (size_t)x->world * MB_SLOT;
  void* p = nullptr;
  // fine-grained: peers write it while this device's kernels poll it - such memory is not held stale in this device's L2.  Where the runtime
  // does not export a fine-grained allocation (hipIpcGetMemHandle fails) the export falls back to plain device memory: the system-scope
  // release / acquire pairs of the two kernels above carry the ordering.
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) {
    x->mbox_fine = true;
  } else {
    (void)hipGetLastError();
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange: out of device memory (mailbox)");
    }
  }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&x->timeout_dev), 4) != hipSuccess || hipMemset(x->timeout_dev, 0, 4) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange: mailbox initialisation");
  }
  x->mbox = static_cast<char*>(p);
  x->peer[x->rank] = x->mbox;
  return EPS_OK;
}

int32_t eps_exchange_create_direct(int32_t rank, int32_t world, int32_t device, eps_exchange** out) {
  if (!out) return EPS_USER_ERROR;
  eps_exchange* x = new eps_exchange();
  *out = x;
  x->rank = rank;
  x->world = world;
  x->device = device;
  if (world < 1 || world > MB_MAX_WORLD || rank < 0 || rank >= world) return x->fail(EPS_USER_ERROR, "eps_exchange_create_direct: rank / world (1..16)");
  for (auto& t : x->ev)
    for (auto& e : t)
      if (hipEventCreate(&e) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_create_direct: hipEventCreate");
  return mbox_alloc(x);
}

int32_t eps_exchange_mailbox_export(eps_exchange* x, void* handle64) {
  if (!x || !handle64) return EPS_USER_ERROR;
  static_assert(sizeof(hipIpcMemHandle_t) == EPS_EXCHANGE_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
  int32_t rc = mbox_alloc(x);
  if (rc != EPS_OK) return rc;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, x->mbox) != hipSuccess) {
    (void)hipGetLastError();
    if (!x->mbox_fine) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_mailbox_export: hipIpcGetMemHandle");
    // (a fine-grained allocation this runtime does not export: once more with plain device memory)
    (void)hipFree(x->mbox);
    x->mbox = nullptr;
    x->mbox_fine = false;
    void* p = nullptr;
    const size_t bytes = MB_HEAD + 2 * (size_t)x->world * MB_SLOT;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_mailbox_export: mailbox in plain device memory");
    }
    x->mbox = static_cast<char*>(p);
    x->peer[x->rank] = x->mbox;
    if (hipIpcGetMemHandle(&h, x->mbox) != hipSuccess) {
      (void)hipGetLastError();
      return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_mailbox_export: hipIpcGetMemHandle");
    }
  }
  std::memcpy(handle64, &h, sizeof h);
  return EPS_OK;
}

int32_t eps_exchange_mailbox_connect(eps_exchange* x, const void* handles) {
  if (!x || !handles) return EPS_USER_ERROR;
  if (!x->mbox) return x->fail(EPS_USER_ERROR, "eps_exchange_mailbox_connect: export this rank's mailbox first");
  if (hipSetDevice(x->device) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
  for (int r = 0; r < x->world; ++r) {
    if (r == x->rank || x->peer[r]) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof h, sizeof h);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return x->fail(EPS_INFRA_UNEXPECTED_ERROR, std::string("eps_exchange_mailbox_connect: hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
    }
    x->peer[r] = static_cast<char*>(p);
  }
  x->connected = true;
  return EPS_OK;
}

int32_t eps_exchange_direct_merge(eps_exchange* x, const int64_t* ids, const float* dist, int64_t nq, int32_t k, int64_t* out_ids, float* out_dist, void* hip_stream) {
  if (!x) return EPS_USER_ERROR;
  if (!x->connected && x->world > 1) return x->fail(EPS_USER_ERROR, "eps_exchange_direct_merge: mailboxes are not connected (eps_exchange_mailbox_export / _connect)");
  if (!ids || !dist || !out_ids || !out_dist || nq < 0 || k <= 0) return x->fail(EPS_USER_ERROR, "eps_exchange_direct_merge: null pointer / nq / k");
  if (nq == 0) return EPS_OK;
  const size_t nk = (size_t)nq * (size_t)k;
  if (nk * 12 > MB_SLOT) return x->fail(EPS_USER_ERROR, "eps_exchange_direct_merge: more than 16 KB of lists per rank - this is the collective's job (eps_exchange_allgather_merge)");
  if (!eps::is_device_ptr(ids) || !eps::is_device_ptr(dist) || !eps::is_device_ptr(out_ids) || !eps::is_device_ptr(out_dist))
    return x->fail(EPS_USER_ERROR, "eps_exchange_direct_merge: lists and results live on the rank's device");
  int32_t rc = mbox_alloc(x);   // (world = 1 without export)
  if (rc != EPS_OK) return rc;
  if (hipSetDevice(x->device) != hipSuccess) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  unsigned told = 0;   // a peer's lists missing in an EARLIER call: said now (the read is a small blocking copy only when the flag word is dirty - it never is on a healthy job)
  if (x->direct_calls % 256 == 255) {
    if (hipMemcpy(&told, x->timeout_dev, 4, hipMemcpyDeviceToHost) == hipSuccess && told) return x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_direct_merge: a peer's lists did not arrive within the wait of an earlier call (rank mask " + std::to_string(told) + ")");
  }
  const unsigned long long call = ++x->direct_calls;
  MboxPeers peers;
  for (int r = 0; r < MB_MAX_WORLD; ++r) peers.p[r] = r < x->world ? x->peer[r] : nullptr;
  hipEvent_t* ev = x->ev[x->calls % eps_exchange::RING];
  (void)hipEventRecord(ev[0], s);
  hipLaunchKernelGGL(mbox_push_kernel, dim3((unsigned)x->world), dim3(256), 0, s, peers, x->rank, x->world, call, ids, dist, (int64_t)nk);
  hipLaunchKernelGGL(mbox_wait_kernel, dim3(1), dim3(64), 0, s, x->mbox, x->world, call, x->timeout_dev, 1000000000ull);   // (100 MHz wall clock: 10 s)
  (void)hipEventRecord(ev[1], s);
  const char* base = x->mbox + MB_HEAD + (call & 1ull) * (size_t)x->world * MB_SLOT;
  eps::launch_merge_shards(reinterpret_cast<const float*>(base + nk * 8), reinterpret_cast<const int64_t*>(base), x->world, nq, k, out_dist, out_ids, s, (int64_t)MB_SLOT);
  (void)hipEventRecord(ev[2], s);
  x->calls += 1;
  return hipGetLastError() == hipSuccess ? EPS_OK : x->fail(EPS_INFRA_UNEXPECTED_ERROR, "eps_exchange_direct_merge: launch");
}

int32_t eps_exchange_times(eps_exchange* x, double* us_pairs, int32_t max_calls) {
  if (!x || !us_pairs || max_calls < 0) return -1;
  const int64_t have = x->calls < eps_exchange::RING ? x->calls : eps_exchange::RING;
  const int64_t want = have < max_calls ? have : max_calls;
  for (int64_t i = 0; i < want; ++i) {   // oldest of the `want` first
    hipEvent_t* ev = x->ev[(x->calls - want + i) % eps_exchange::RING];
    float a = 0.f, b = 0.f;
    if (hipEventSynchronize(ev[2]) != hipSuccess || hipEventElapsedTime(&a, ev[0], ev[1]) != hipSuccess || hipEventElapsedTime(&b, ev[1], ev[2]) != hipSuccess) {
      x->err = "eps_exchange_times: event read-back";
      return -1;
    }
    us_pairs[2 * i] = 1e3 * (double)a;
    us_pairs[2 * i + 1] = 1e3 * (double)b;
  }
  return (int32_t)want;
}

int32_t eps_exchange_info(eps_exchange* x, int32_t* rank, int32_t* world, int32_t* rccl_version, char* rccl_path, int64_t cap) {
  if (!x) return EPS_USER_ERROR;
  Rccl& r = rccl();
  if (rank) *rank = x->rank;
  if (world) *world = x->world;
  if (rccl_version) *rccl_version = r.version;
  if (rccl_path && cap > 0) {
    std::strncpy(rccl_path, r.path.c_str(), (size_t)cap - 1);
    rccl_path[cap - 1] = 0;
  }
  return EPS_OK;
}

const char* eps_exchange_last_error(eps_exchange* x) { return x ? x->err.c_str() : "null exchange"; }

void eps_exchange_destroy(eps_exchange* x) {
  if (!x) return;
  if (x->comm) {
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    (void)rccl().CommDestroy(x->comm);
  }
  if (x->mbox || x->timeout_dev) {
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; ++r)
      if (r != x->rank && x->peer[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
    if (x->mbox) (void)hipFree(x->mbox);
    if (x->timeout_dev) (void)hipFree(x->timeout_dev);
  }
  for (auto& t : x->ev)
    for (auto& e : t)
      if (e) (void)hipEventDestroy(e);
  delete x;
}

}  // extern "C"
