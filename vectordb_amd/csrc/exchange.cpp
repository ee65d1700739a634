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

struct eps_exchange {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
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
  for (auto& t : x->ev)
    for (auto& e : t)
      if (e) (void)hipEventDestroy(e);
  delete x;
}

}  // extern "C"
