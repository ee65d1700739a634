// Host side of libepsilla_gfx950 — see index.hpp.  Compiled with hipcc (host code only here).
#include "index.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <unordered_set>

namespace eps {

// ------------------------------------------------------------------------------------------------ engine-selection switches
// (eps_set_tuning).  Values are interned and never freed, so a pointer handed out stays valid while another thread replaces the entry.
namespace {
std::mutex g_tune_mu;
std::unordered_map<std::string, const char*> g_tune;
std::unordered_set<std::string> g_tune_values;   // (interned: a value set a million times is stored once; element addresses are stable)
}  // namespace
const char* tune_env(const char* name) {
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (!g_tune.empty()) {
      auto it = g_tune.find(name);
      if (it != g_tune.end()) return it->second;
    }
  }
#ifdef EPS_LAB
  return getenv(name);
#else
  return nullptr;
#endif
}
int tune_int(const char* name, int dflt) {
  const char* e = tune_env(name);
  return e ? atoi(e) : dflt;
}
static void tune_set(const char* name, const char* value) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  if (!name) {
    g_tune.clear();
  } else if (!value) {
    g_tune.erase(name);
  } else {
    g_tune[name] = g_tune_values.emplace(value).first->c_str();
  }
}

// ------------------------------------------------------------------------------------------------ utils
DevBuf::~DevBuf() { release(); }
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}
Index::HostBuf::~HostBuf() {
  if (p) (void)hipHostFree(p);
}
bool Index::HostBuf::reserve(size_t bytes) {
  if (bytes <= cap) return true;
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    p = nullptr;
    return false;
  }
  cap = want;
  return true;
}
namespace {
std::mutex g_scratch_mu;
std::vector<ScratchClaim*> g_scratch;
}  // namespace
ScratchClaim::ScratchClaim() {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  g_scratch.push_back(this);
}
ScratchClaim::~ScratchClaim() {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  g_scratch.erase(std::remove(g_scratch.begin(), g_scratch.end(), this), g_scratch.end());
}
size_t scratch_reclaim() {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  size_t freed = 0;
  for (ScratchClaim* c : g_scratch) {
    if (!c->buf || !c->buf->p) continue;
    if (!c->try_enter()) continue;   // its owner is between "the table is there" and the launch that uses it (possibly this very thread)
    freed += c->buf->cap;
    c->buf->release();
    c->leave();
  }
  return freed;
}
bool DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return true;
  release();
  for (int attempt = 0; attempt < 2; ++attempt) {
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) == hipSuccess) {
      cap = want;
      return true;
    }
    (void)hipGetLastError();
    p = nullptr;
    if (hipMalloc(&p, bytes) == hipSuccess) {
      cap = bytes;
      return true;
    }
    (void)hipGetLastError();
    p = nullptr;
    if (attempt == 0 && scratch_reclaim() == 0) break;   // nothing to give back: fail; else once more
  }
  return false;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

Index::Index(int64_t dim, int metric, int device) : dim_(dim), metric_(metric), device_(device) {}

Index::~Index() {
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (graph_) graph_free(graph_);
  if (mirror_) half_mirror_free(mirror_);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  for (auto& pr : kring_)
    for (auto& e : pr)
      if (e) (void)hipEventDestroy(e);
  for (auto& pr : stage_ev_)
    for (auto& e : pr)
      if (e) (void)hipEventDestroy(e);
  if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

int32_t Index::hip_fail(hipError_t e, const char* what) {
  (void)hipGetLastError();
  return fail(EPS_INFRA_UNEXPECTED_ERROR, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e__ = (expr);                           \
    if (e__ != hipSuccess) return hip_fail(e__, #expr); \
  } while (0)

int32_t Index::init() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return fail(EPS_INFRA_UNEXPECTED_ERROR, "no HIP device available: libepsilla_gfx950 has no CPU fallback");
  }
  if (device_ < 0 || device_ >= count) return fail(EPS_USER_ERROR, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device_));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(EPS_INFRA_UNEXPECTED_ERROR, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  own_stream_ = true;
  HIP_TRY(hipEventCreate(&ev0_));
  HIP_TRY(hipEventCreate(&ev1_));
  for (auto& pr : kring_)
    for (auto& e : pr) HIP_TRY(hipEventCreate(&e));
  evk0_ = kring_[0][0];
  evk1_ = kring_[0][1];
  for (auto& pr : stage_ev_)
    for (auto& e : pr) HIP_TRY(hipEventCreate(&e));
  return EPS_OK;
}

int32_t Index::set_stream(void* s) {
  HIP_TRY(hipSetDevice(device_));
  if (stream_) HIP_TRY(hipStreamSynchronize(stream_));
  if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
  own_stream_ = false;
  if (s) {
    stream_ = static_cast<hipStream_t>(s);
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    own_stream_ = true;
  }
  return EPS_OK;
}

int32_t Index::synchronize() {
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamSynchronize(stream_));
  return EPS_OK;
}

int32_t Index::attach_rows(const float* rows, int64_t n) { return attach_rows_strided(rows, n, dim_); }
int32_t Index::append_rows(const float* rows, int64_t n_new) { return append_rows_strided(rows, n_new, dim_); }

int32_t Index::clone_rows(IndexBase& src_base, int64_t n) {
  Index* src = dynamic_cast<Index*>(&src_base);
  if (!src || src == this) return fail(EPS_USER_ERROR, "clone_rows: the source must be another plain index");
  if (src->dim_ != dim_ || src->device_ != device_) return fail(EPS_USER_ERROR, "clone_rows: source and destination differ in dimension or device");
  if (n < 0 || n > src->n_rows_) return fail(EPS_USER_ERROR, "clone_rows: n exceeds the source's rows");
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamSynchronize(src->stream_));   // (the source's rows are complete)
  return attach_rows_strided(src->d_rows_, n, dim_, true);
}

int32_t Index::attach_rows_strided(const float* rows, int64_t n, int64_t pitch, bool copy_device_rows) {
  if (n < 0 || (n > 0 && !rows) || pitch < dim_) return fail(EPS_USER_ERROR, "attach_rows: bad arguments");
  if (n >= (int64_t)1 << 31) return fail(EPS_DB_UNSUPPORTED_ERROR, "attach_rows: more than 2^31-1 rows per index (shard first)");
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamSynchronize(stream_));
  if (is_device_ptr(rows) && pitch == dim_ && !copy_device_rows) {
    rows_buf_.release();
    d_rows_ = rows;
    rows_owned_ = false;
  } else {
    const size_t bytes = (size_t)n * dim_ * sizeof(float);
    if (!rows_buf_.reserve(bytes ? bytes : 16)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "attach_rows: out of device memory");
    if (bytes && pitch == dim_) HIP_TRY(hipMemcpyAsync(rows_buf_.p, rows, bytes, hipMemcpyDefault, stream_));
    else if (bytes) HIP_TRY(hipMemcpy2DAsync(rows_buf_.p, (size_t)dim_ * 4, rows, (size_t)pitch * 4, (size_t)dim_ * 4, (size_t)n, hipMemcpyDefault, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    d_rows_ = rows_buf_.as<float>();
    rows_owned_ = true;
  }
  n_rows_ = n;
  ++rows_version_;
  prog_rows_host_ = nullptr;   // a new table: attribute rows cached from the previous one are not its rows, whatever their address
  prog_rows_uploaded_ = 0;
  loaded_attr_rows_ = 0;       // ... nor are the rows a previous eps_index_load_table kept (load_table sets them again after attaching)
  // state that was sized for the previous table does not carry over: a graph over more rows than are attached now, a
  // deleted bitset or an attribute column of the old length
  if (n_indexed_ > n) {
    n_indexed_ = 0;
    nav_ = 0;
    h_off_.assign(1, 0);
    h_nbr_.clear();
    (void)graph_upload(*this);
  }
  if (deleted_bytes_ < (n + 7) / 8) {
    d_deleted_ = nullptr;
    deleted_bytes_ = 0;
  }
  if (fcol_rows_ < n) {
    f_op_ = 0;
    d_fcol_ = nullptr;
    fcol_rows_ = 0;
  }
  return EPS_OK;
}

int32_t Index::append_rows_strided(const float* rows, int64_t n_new, int64_t pitch) {
  if (n_new < 0 || (n_new > 0 && !rows) || pitch < dim_) return fail(EPS_USER_ERROR, "append_rows: bad arguments");
  if (n_new == 0) return EPS_OK;
  if (n_rows_ > 0 && !rows_owned_) return fail(EPS_USER_ERROR, "append_rows: the row store is borrowed device memory; re-attach instead");
  HIP_TRY(hipSetDevice(device_));
  HIP_TRY(hipStreamSynchronize(stream_));
  const size_t old_bytes = (size_t)n_rows_ * dim_ * sizeof(float);
  const size_t add_bytes = (size_t)n_new * dim_ * sizeof(float);
  if (old_bytes + add_bytes > rows_buf_.cap) {
    DevBuf bigger;
    if (!bigger.reserve((old_bytes + add_bytes) * 3 / 2)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "append_rows: out of device memory");
    if (old_bytes) HIP_TRY(hipMemcpyAsync(bigger.p, rows_buf_.p, old_bytes, hipMemcpyDeviceToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    rows_buf_.release();
    rows_buf_.p = bigger.p;
    rows_buf_.cap = bigger.cap;
    bigger.p = nullptr;
    bigger.cap = 0;
  }
  if (pitch == dim_) HIP_TRY(hipMemcpyAsync(static_cast<char*>(rows_buf_.p) + old_bytes, rows, add_bytes, hipMemcpyDefault, stream_));
  else HIP_TRY(hipMemcpy2DAsync(static_cast<char*>(rows_buf_.p) + old_bytes, (size_t)dim_ * 4, rows, (size_t)pitch * 4, (size_t)dim_ * 4, (size_t)n_new,
                                hipMemcpyDefault, stream_));
  HIP_TRY(hipStreamSynchronize(stream_));
  d_rows_ = rows_buf_.as<float>();
  rows_owned_ = true;
  n_rows_ += n_new;   // rows_version_ stays: the fp16 mirror is extended by the new rows on the next MFMA search
  return EPS_OK;   // (a bitset / attribute column that is now too short is rejected by search(), see there)
}

// data_mvp.bin -> HBM (layout: db/table_segment_mvp.cpp:939-1010; the reference's own loader is its constructor, :133-295)
int32_t Index::load_table(const char* path, const eps_table_layout* lay, int64_t* n_out) {
  if (!path || !lay || lay->primitive_offset < 0 || lay->var_len_attrs < 0 || lay->dense_fields <= 0 || !lay->dense_dims || lay->field < 0 ||
      lay->field >= lay->dense_fields)
    return fail(EPS_USER_ERROR, "load_table: bad arguments");
  if (lay->dense_dims[lay->field] != dim_) return fail(EPS_USER_ERROR, "load_table: the field's dimension differs from the index's");
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Cannot open file: ") + path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 32) {
    ::close(fd);
    return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Corrupt table segment file: ") + path);
  }
  const size_t fsize = (size_t)st.st_size;
  void* map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (map == MAP_FAILED) return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Cannot map file: ") + path);
  struct Unmap {
    void* p;
    size_t n;
    ~Unmap() { munmap(p, n); }
  } unmap{map, fsize};
  const char* base = static_cast<const char*>(map);
  size_t pos = 0;
  auto need = [&](size_t bytes) { return bytes <= fsize && pos <= fsize - bytes; };
  auto corrupt = [&]() { return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Corrupt table segment file: ") + path); };
  if (!need(24)) return corrupt();
  uint64_t n64;
  int64_t first_id, bitset_size;
  std::memcpy(&n64, base + pos, 8);
  std::memcpy(&first_id, base + pos + 8, 8);
  std::memcpy(&bitset_size, base + pos + 16, 8);
  pos += 24;
  (void)first_id;
  if (n64 >= ((uint64_t)1 << 31) || bitset_size < 0 || !need((size_t)bitset_size)) return corrupt();
  const int64_t n = (int64_t)n64;
  if (bitset_size < (n + 7) / 8) return corrupt();   // the reference writes the whole ConcurrentBitset (capacity bits >= record count)
  const uint8_t* bits = reinterpret_cast<const uint8_t*>(base + pos);
  pos += (size_t)bitset_size;
  if ((uint64_t)lay->primitive_offset > fsize || (n > 0 && (uint64_t)lay->primitive_offset > (uint64_t)fsize / (uint64_t)n)) return corrupt();   // (n * offset cannot wrap)
  const size_t attr_bytes = (size_t)n * (size_t)lay->primitive_offset;
  if (!need(attr_bytes)) return corrupt();
  const char* attrs = base + pos;
  pos += attr_bytes;
  for (int64_t r = 0; r < n; ++r)            // variable-length attributes: int64 length + payload each
    for (int a = 0; a < lay->var_len_attrs; ++a) {
      if (!need(8)) return corrupt();
      int64_t len;
      std::memcpy(&len, base + pos, 8);
      pos += 8;
      if (len < 0 || !need((size_t)len)) return corrupt();
      pos += (size_t)len;
    }
  const float* field_rows = nullptr;
  for (int f = 0; f < lay->dense_fields; ++f) {
    if (lay->dense_dims[f] <= 0) return fail(EPS_USER_ERROR, "load_table: bad dimension");
    const size_t bytes = (size_t)n * (size_t)lay->dense_dims[f] * sizeof(float);
    if (!need(bytes)) return corrupt();
    if (f == lay->field) field_rows = reinterpret_cast<const float*>(base + pos);
    pos += bytes;
  }
  if (!need(8)) return corrupt();            // trailing WAL id
  int32_t rc = attach_rows(field_rows, n);   // the mapped pages go to HBM directly; no host copy of the table is made
  if (rc != EPS_OK) return rc;
  rc = n > 0 ? set_deleted(bits, bitset_size) : set_deleted(nullptr, 0);   // (never a bitset left over from a previous table)
  if (rc != EPS_OK) return rc;
  if (attr_bytes > 0) {   // keep the attribute rows on the device for a later filter program
    HIP_TRY(hipStreamSynchronize(stream_));
    if (!prog_rows_buf_.reserve(attr_bytes)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "load_table: out of device memory (attribute rows)");
    HIP_TRY(hipMemcpyAsync(prog_rows_buf_.p, attrs, attr_bytes, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    d_prog_rows_ = prog_rows_buf_.as<uint8_t>();
    prog_rows_host_ = nullptr;
    prog_rows_stride_ = lay->primitive_offset;
    prog_rows_uploaded_ = n;
    loaded_attr_rows_ = n;
    loaded_attr_stride_ = lay->primitive_offset;
    prog_len_ = 0;
  }
  if (n_out) *n_out = n;
  return EPS_OK;
}

int32_t Index::set_id_map(int64_t base, int64_t stride) {
  if (stride <= 0) return fail(EPS_USER_ERROR, "set_id_map: stride must be positive");
  id_base_ = base;
  id_stride_ = stride;
  return EPS_OK;
}

int32_t Index::set_deleted(const uint8_t* bits, int64_t nbytes) {
  HIP_TRY(hipSetDevice(device_));
  if (!bits || nbytes <= 0) {
    d_deleted_ = nullptr;
    deleted_bytes_ = 0;
    return EPS_OK;
  }
  if (nbytes < (n_rows_ + 7) / 8) return fail(EPS_USER_ERROR, "set_deleted: bitset shorter than ceil(rows/8) bytes");
  deleted_bytes_ = nbytes;
  if (is_device_ptr(bits)) {
    d_deleted_ = bits;
  } else {
    // the DBMS hands over its bitset on every call; a table without deletions (the common case) is recognised here, which
    // skips the upload and keeps the unfiltered fast paths (MFMA-seeded staging) available
    const int64_t live = (n_rows_ + 7) / 8;
    bool any = false;
    int64_t i = 0;
    for (; i + 8 <= live && !any; i += 8) {
      uint64_t w;
      std::memcpy(&w, bits + i, 8);
      any = w != 0;
    }
    for (; i < live && !any; ++i) any = bits[i] != 0;
    if (!any && n_rows_ > 0) {   // (before any rows are attached nothing can be concluded from the live prefix)
      d_deleted_ = nullptr;
      return EPS_OK;
    }
    if (!deleted_buf_.reserve((size_t)nbytes)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "set_deleted: out of device memory");
    HIP_TRY(hipMemcpyAsync(deleted_buf_.p, bits, (size_t)nbytes, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));  // the host bitset may change right after we return
    d_deleted_ = deleted_buf_.as<uint8_t>();
  }
  return EPS_OK;
}

int32_t Index::set_int_filter(const void* column, int64_t stride, int32_t width, int32_t op, int64_t constant) {
  HIP_TRY(hipSetDevice(device_));
  if (op == EPS_OP_NONE || !column) {
    f_op_ = 0;
    d_fcol_ = nullptr;
    fcol_rows_ = 0;
    return EPS_OK;
  }
  if (op < 0 || op > EPS_OP_NE) return fail(EPS_USER_ERROR, "set_int_filter: unknown operator");
  if (width != 1 && width != 2 && width != 4 && width != 8) return fail(EPS_USER_ERROR, "set_int_filter: width must be 1, 2, 4 or 8 bytes");
  if (stride < width) return fail(EPS_USER_ERROR, "set_int_filter: stride smaller than the value width");
  if (is_device_ptr(column)) {
    d_fcol_ = static_cast<const uint8_t*>(column);
  } else {
    const size_t bytes = (size_t)(n_rows_ > 0 ? (n_rows_ - 1) * stride + width : 0);
    if (!fcol_buf_.reserve(bytes ? bytes : 16)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "set_int_filter: out of device memory");
    if (bytes) HIP_TRY(hipMemcpyAsync(fcol_buf_.p, column, bytes, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    d_fcol_ = fcol_buf_.as<uint8_t>();
  }
  f_stride_ = stride;
  f_width_ = width;
  f_op_ = op;
  f_value_ = constant;
  fcol_rows_ = n_rows_;
  return EPS_OK;
}

int32_t Index::set_filter_program(const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride, int64_t n_rows, int32_t flags) {
  return set_filter_program_pitched(ops, nops, rows, stride, stride, n_rows, flags);
}

// rows: row i at rows + i * src_pitch, row_bytes (<= src_pitch) of it are the attribute row.  A hash-sharded table hands every
// shard the same memory with src_pitch = shards * row_bytes: only the shard's own rows are uploaded, packed at row_bytes.
int32_t Index::set_filter_program_pitched(const eps_filter_op* ops, int32_t nops, const void* rows, int64_t src_pitch, int64_t row_bytes, int64_t n_rows,
                                          int32_t flags) {
  HIP_TRY(hipSetDevice(device_));
  if (nops <= 0 || !ops) {
    prog_len_ = 0;
    d_prog_rows_ = nullptr;
    prog_uses_dist_ = false;
    return EPS_OK;
  }
  if (nops > 64) return fail(EPS_DB_UNSUPPORTED_ERROR, "set_filter_program: more than 64 instructions");
  const bool use_loaded = !rows && loaded_attr_rows_ > 0 && loaded_attr_rows_ >= n_rows_;   // the rows eps_index_load_table kept
  int64_t stride = row_bytes;   // of the device copy
  if (use_loaded) {
    stride = loaded_attr_stride_;
    n_rows = loaded_attr_rows_;
  }
  if ((!rows && !use_loaded) || stride <= 0 || n_rows < n_rows_ || (!use_loaded && src_pitch < row_bytes))
    return fail(EPS_USER_ERROR, "set_filter_program: attribute rows missing or shorter than the table");
  // validate: known opcodes, attribute loads inside a row, stack discipline
  int sp = 0, maxsp = 0;
  bool uses_dist = false;
  for (int i = 0; i < nops; ++i) {
    const int op = ops[i].op;
    if (op < EPS_FOP_PUSH_CONST || op > EPS_FOP_NE_BOOL) return fail(EPS_USER_ERROR, "set_filter_program: unknown opcode");
    if (op <= EPS_FOP_PUSH_BOOL) {
      static const int width[] = {0, 0, 0, 1, 2, 4, 8, 4, 8, 1};
      if (op >= EPS_FOP_PUSH_I8 && (ops[i].arg < 0 || ops[i].arg + width[op] > stride))
        return fail(EPS_USER_ERROR, "set_filter_program: attribute offset outside the row");
      uses_dist |= op == EPS_FOP_PUSH_DIST;
      ++sp;
    } else if (op == EPS_FOP_NOT) {
      if (sp < 1) return fail(EPS_USER_ERROR, "set_filter_program: stack underflow");
    } else {
      if (sp < 2) return fail(EPS_USER_ERROR, "set_filter_program: stack underflow");
      --sp;
    }
    maxsp = std::max(maxsp, sp);
  }
  if (sp != 1 || maxsp > 16) return fail(EPS_USER_ERROR, "set_filter_program: the program must leave exactly one value (stack depth <= 16)");
  HIP_TRY(hipStreamSynchronize(stream_));
  if (!prog_buf_.reserve((size_t)nops * sizeof(FilterOp))) return fail(EPS_INFRA_UNEXPECTED_ERROR, "set_filter_program: out of device memory");
  static_assert(sizeof(FilterOp) == sizeof(eps_filter_op), "FilterOp mirrors eps_filter_op");
  HIP_TRY(hipMemcpyAsync(prog_buf_.p, ops, (size_t)nops * sizeof(FilterOp), hipMemcpyHostToDevice, stream_));
  if (use_loaded) {
    d_prog_rows_ = prog_rows_buf_.as<uint8_t>();
  } else if (is_device_ptr(rows)) {
    d_prog_rows_ = static_cast<const uint8_t*>(rows);
    stride = src_pitch;   // used in place
    prog_rows_host_ = nullptr;
    loaded_attr_rows_ = 0;
  } else {
    loaded_attr_rows_ = 0;
    // Default: the whole table is uploaded (the caller may have edited it in place, or another table may live at the same address).
    // EPS_FILTER_ROWS_APPEND_ONLY is the caller's promise that rows handed over earlier FROM THIS POINTER are unchanged - true for
    // the reference's attribute table, where an update is delete + insert (table_segment_mvp.cpp:476-587) - then only the new
    // tail crosses PCIe.
    const size_t bytes = (size_t)n_rows * (size_t)stride;
    const bool same = (flags & EPS_FILTER_ROWS_APPEND_ONLY) && prog_rows_host_ == rows && prog_rows_stride_ == stride && prog_rows_pitch_ == src_pitch &&
                      prog_rows_uploaded_ <= n_rows && prog_rows_buf_.p;
    const int64_t have_rows = same ? prog_rows_uploaded_ : 0;
    const size_t have = (size_t)have_rows * (size_t)stride;
    if (bytes > prog_rows_buf_.cap) {
      DevBuf bigger;
      if (!bigger.reserve(bytes + bytes / 2 + 16)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "set_filter_program: out of device memory");
      if (have) HIP_TRY(hipMemcpyAsync(bigger.p, prog_rows_buf_.p, have, hipMemcpyDeviceToDevice, stream_));
      HIP_TRY(hipStreamSynchronize(stream_));
      prog_rows_buf_.release();
      prog_rows_buf_.p = bigger.p;
      prog_rows_buf_.cap = bigger.cap;
      bigger.p = nullptr;
      bigger.cap = 0;
    }
    if (n_rows > have_rows) {
      const char* src = static_cast<const char*>(rows) + (size_t)have_rows * (size_t)src_pitch;
      char* dst = static_cast<char*>(prog_rows_buf_.p) + have;
      if (src_pitch == stride) HIP_TRY(hipMemcpyAsync(dst, src, bytes - have, hipMemcpyHostToDevice, stream_));
      else HIP_TRY(hipMemcpy2DAsync(dst, (size_t)stride, src, (size_t)src_pitch, (size_t)stride, (size_t)(n_rows - have_rows), hipMemcpyHostToDevice, stream_));
    }
    d_prog_rows_ = prog_rows_buf_.as<uint8_t>();
    prog_rows_host_ = rows;
    prog_rows_stride_ = stride;
    prog_rows_pitch_ = src_pitch;
    prog_rows_uploaded_ = n_rows;
  }
  HIP_TRY(hipStreamSynchronize(stream_));
  prog_stride_ = stride;
  prog_rows_n_ = n_rows;
  prog_len_ = nops;
  prog_uses_dist_ = uses_dist;
  f_op_ = 0;   // replaces the single-comparison filter
  d_fcol_ = nullptr;
  fcol_rows_ = 0;
  return EPS_OK;
}

FilterSpec Index::filter_spec() const {
  FilterSpec f = no_filter();
  f.deleted = d_deleted_;
  f.column = f_op_ ? d_fcol_ : nullptr;
  f.stride = f_stride_;
  f.width = f_width_;
  f.op = f_op_;
  f.value = f_value_;
  if (prog_len_ > 0 && d_prog_rows_) {
    f.prog = prog_buf_.as<FilterOp>();
    f.prog_rows = d_prog_rows_;
    f.prog_stride = prog_stride_;
    f.prog_len = prog_len_;
    f.prog_use_dist = prefilter_call_ ? 0 : 1;   // PreFilterBruteForceSearch evaluates the filter without a distance (:795)
  }
  return f;
}

// ------------------------------------------------------------------------------------------------ graph
int32_t Index::set_graph(int64_t n, const int64_t* off, const int64_t* nbr, int64_t nav) {
  if (n < 0 || (n > 0 && (!off || (!nbr && off[n] > 0)))) return fail(EPS_USER_ERROR, "set_graph: bad arguments");
  if (n > n_rows_) return fail(EPS_USER_ERROR, "set_graph: graph has more nodes than attached rows");
  if (n > 0 && (nav < 0 || nav >= n)) return fail(EPS_USER_ERROR, "set_graph: navigation point out of range");
  if (n > 0) {
    if (off[0] != 0) return fail(EPS_USER_ERROR, "set_graph: offsets[0] must be 0");
    for (int64_t i = 0; i < n; ++i)
      if (off[i + 1] < off[i]) return fail(EPS_USER_ERROR, "set_graph: offsets must be non-decreasing");
    const int64_t e = off[n];
    for (int64_t i = 0; i < e; ++i)
      if (nbr[i] < 0 || nbr[i] >= n) return fail(EPS_USER_ERROR, "set_graph: neighbor id out of range");
    h_off_.assign(off, off + n + 1);
    h_nbr_.assign(nbr, nbr + e);
  } else {
    h_off_.assign(1, 0);
    h_nbr_.clear();
  }
  n_indexed_ = n;
  nav_ = nav;
  return graph_upload(*this);
}

int32_t Index::graph_info(int64_t* n, int64_t* edges, int64_t* nav) const {
  if (n) *n = n_indexed_;
  if (edges) *edges = n_indexed_ > 0 ? h_off_[n_indexed_] : 0;
  if (nav) *nav = nav_;
  return EPS_OK;
}

int32_t Index::get_graph(int64_t* off, int64_t* nbr) const {
  if (!off || !nbr) return EPS_USER_ERROR;
  if (h_off_.empty()) {
    off[0] = 0;
    return EPS_OK;
  }
  std::memcpy(off, h_off_.data(), sizeof(int64_t) * h_off_.size());
  if (!h_nbr_.empty()) std::memcpy(nbr, h_nbr_.data(), sizeof(int64_t) * h_nbr_.size());
  return EPS_OK;
}

// ann_graph_<field>.bin, byte-compatible with ANNGraphSegment::SaveANNGraph (db/ann_graph_segment.cpp:156-199):
// i64 n, i64 first_record_id, i64 offsets[n+1], i64 neighbors[E], i64 navigation_point; tmp + fsync + rename.
int32_t Index::save_graph(const char* path) {
  if (!path) return fail(EPS_USER_ERROR, "save_graph: null path");
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Cannot open file: ") + path);
  const int64_t n = n_indexed_, first = 0;
  std::vector<int64_t> zero(1, 0);
  const int64_t* off = h_off_.empty() ? zero.data() : h_off_.data();
  const int64_t e = off[n];
  bool ok = std::fwrite(&n, 8, 1, f) == 1 && std::fwrite(&first, 8, 1, f) == 1 &&
            std::fwrite(off, 8, (size_t)n + 1, f) == (size_t)n + 1 &&
            (e == 0 || std::fwrite(h_nbr_.data(), 8, (size_t)e, f) == (size_t)e) && std::fwrite(&nav_, 8, 1, f) == 1;
  std::fflush(f);
  fsync(fileno(f));
  std::fclose(f);
  if (!ok) return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Failed to write to file: ") + path);
  if (std::rename(tmp.c_str(), path) != 0)
    return fail(EPS_INFRA_UNEXPECTED_ERROR, "Failed to rename temp file: " + tmp + " to " + path);
  return EPS_OK;
}

int32_t Index::load_graph(const char* path) {
  if (!path) return fail(EPS_USER_ERROR, "load_graph: null path");
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Cannot open file: ") + path);
  int64_t hdr[2];
  std::vector<int64_t> off, nbr;
  int64_t nav = 0;
  std::fseek(f, 0, SEEK_END);
  const int64_t fsize = (int64_t)std::ftell(f);   // header values are untrusted: bound them by the file size
  std::fseek(f, 0, SEEK_SET);
  bool ok = std::fread(hdr, 8, 2, f) == 2 && hdr[0] >= 0 && hdr[0] <= (fsize - 32) / 8;
  if (ok) {
    off.resize((size_t)hdr[0] + 1);
    ok = std::fread(off.data(), 8, off.size(), f) == off.size() && off[hdr[0]] >= 0 && off[hdr[0]] <= (fsize - 24 - 8 * (hdr[0] + 1)) / 8;
  }
  if (ok) {
    nbr.resize((size_t)off[hdr[0]]);
    ok = (nbr.empty() || std::fread(nbr.data(), 8, nbr.size(), f) == nbr.size()) && std::fread(&nav, 8, 1, f) == 1;
  }
  std::fclose(f);
  if (!ok) return fail(EPS_DB_UNEXPECTED_ERROR, std::string("Corrupt ANN graph file: ") + path);
  return set_graph(hdr[0], off.data(), nbr.data(), nav);
}

int32_t Index::build(int64_t n, const eps_build_params* p) {
  eps_build_params bp;
  if (p) bp = *p; else eps_default_build_params(&bp);
  if (n < 0 || n > n_rows_) return fail(EPS_USER_ERROR, "build: n exceeds the attached rows");
  HIP_TRY(hipSetDevice(device_));
  return graph_build(*this, n, bp);
}

// ------------------------------------------------------------------------------------------------ search
int32_t Index::flat_stream(const float* dq, int64_t nq, int k, int64_t row_begin, int64_t row_end, u64* run_keys,
                           bool merge_run, int metric, bool filtered) {
  if (k <= 1024) return flat_stream_page(dq, nq, k, row_begin, row_end, run_keys, merge_run, metric, filtered, nullptr, 0);
  // More than 1024 results per query (the reference's BruteForceSearch has no cap: it sorts all n candidates,
  // vec_search_executor.cpp:756-767): pages of 1024 - page p is the scan's 1024 best keys ordered AFTER the last key of page
  // p-1 ((dist, id) keys are unique per row, so the pages are disjoint and their concatenation is the sorted answer).
  if (merge_run) return fail(EPS_DB_UNSUPPORTED_ERROR, "search: merging into an existing result list of more than 1024 entries is not supported", EPS_ERRCLASS_DEVICE_RANGE);
  if (!page_buf_.reserve((size_t)nq * 1024 * sizeof(u64))) return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (result page)");
  for (int done = 0; done < k; done += 1024) {
    const int kc = std::min(1024, k - done);
    const int32_t rc = flat_stream_page(dq, nq, kc, row_begin, row_end, page_buf_.as<u64>(), false, metric, filtered,
                                        done ? run_keys + (done - 1) : nullptr, k);
    if (rc != EPS_OK) return rc;
    HIP_TRY(hipMemcpy2DAsync(run_keys + done, (size_t)k * sizeof(u64), page_buf_.p, (size_t)kc * sizeof(u64), (size_t)kc * sizeof(u64), (size_t)nq,
                             hipMemcpyDeviceToDevice, stream_));
  }
  return EPS_OK;
}

int32_t Index::flat_stream_page(const float* dq, int64_t nq, int k, int64_t row_begin, int64_t row_end, u64* run_keys,
                                bool merge_run, int metric, bool filtered, const u64* lo, int64_t lo_stride) {
  if (row_end <= row_begin) {
    if (!merge_run) launch_fill_u64(run_keys, nq * k, KEY_EMPTY, stream_);
    return EPS_OK;
  }
  const int W = flat_scan_waves(row_end - row_begin, nq, (int)dim_);
  if (!partial_buf_.reserve((size_t)nq * W * k * sizeof(u64))) return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (partial lists)");
  FlatScanArgs a;
  a.rows = d_rows_;
  a.row_begin = row_begin;
  a.row_end = row_end;
  a.dim = (int)dim_;
  a.metric = metric < 0 ? metric_ : metric;
  a.queries = dq;
  a.nq = nq;
  a.k = k;
  a.f = filter_spec();
  if (!filtered) a.f = no_filter();
  a.partial = partial_buf_.as<u64>();
  a.W = W;
  a.thr_in = nullptr;
  a.lo_in = lo;
  a.lo_stride = lo_stride;
  HIP_TRY(hipEventRecord(evk0_, stream_));
  launch_flat_scan(a, stream_);
  HIP_TRY(hipEventRecord(evk1_, stream_));
  launch_merge_lists(a.partial, W * k, k, nq, run_keys, merge_run, stream_);
  HIP_TRY(hipGetLastError());
  stats_.main_kernel_launches += 1;
  stats_.main_kernel_rows = row_end - row_begin;
  stats_.main_kernel_queries = nq;
  stats_.main_kernel_bits = 32;
  stats_.dist_evals += nq * (row_end - row_begin);
  return EPS_OK;
}

int32_t Index::search(const float* queries, int64_t nq, int32_t k, const eps_search_params* pp, int64_t* ids,
                      float* dist, int32_t* counts, int32_t walk_limit) {
  eps_search_params p;
  if (pp) p = *pp; else eps_default_search_params(&p);
  walk_limit_ = walk_limit;
  prefilter_call_ = p.prefilter != 0;
  if (nq < 0 || k <= 0) return fail(EPS_USER_ERROR, "search: nq must be >= 0 and k > 0");
  if (nq == 0) return EPS_OK;
  if (!queries || !ids || !dist) return fail(EPS_USER_ERROR, "search: null buffer");
  if (k > (1 << 20)) return fail(EPS_DB_UNSUPPORTED_ERROR, "search: k > 1048576 is not supported");
  if (p.master_queue <= 0 || p.local_queue <= 0 || p.sync_interval <= 0 || p.intra_threads <= 0)
    return fail(EPS_USER_ERROR, "search: queue sizes, sync interval and thread count must be positive");
  HIP_TRY(hipSetDevice(device_));
  std::memset(&stats_, 0, sizeof(stats_));
  stage_n_ = 0;
  if (d_deleted_ && deleted_bytes_ < (n_rows_ + 7) / 8)
    return fail(EPS_USER_ERROR, "search: the deleted bitset is shorter than the table (rows were appended): call set_deleted again");
  if (f_op_ && d_fcol_ && fcol_rows_ < n_rows_)
    return fail(EPS_USER_ERROR, "search: the filter column is shorter than the table (rows were appended): call set_int_filter again");
  if (prog_len_ > 0 && prog_rows_n_ < n_rows_)
    return fail(EPS_USER_ERROR, "search: the filter program's attribute rows are shorter than the table (rows were appended): call set_filter_program again");
  kring_seq_ += 1;
  {
    const int slot = (int)(kring_seq_ % KRING);
    evk0_ = kring_[slot][0];
    evk1_ = kring_[slot][1];
    kring_valid_[slot] = false;
  }

  const bool q_dev = is_device_ptr(queries);
  const bool out_dev = is_device_ptr(ids);
  if (out_dev != is_device_ptr(dist) || (counts && out_dev != is_device_ptr(counts)))
    return fail(EPS_USER_ERROR, "search: ids_out, dist_out and counts_out must all be host or all be device pointers");

  const float* dq = queries;
  if (!q_dev) {
    const size_t qb = (size_t)nq * dim_ * sizeof(float);
    if (!q_buf_.reserve(qb)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (queries)");
    const bool staging = !(tune_int("EPS_HOST_STAGING", 1) == 0);   // (A/B switch)
    if (staging && qb <= ((size_t)256 << 10) && h_q_.reserve(qb)) {   // (a few vectors: -30 us per call; a 3 MB batch: the runtime's pageable path measured faster than memcpy + DMA)
      // (the previous call's copy out of h_q_ has completed: every call with host queries ends in a stream sync or its results are device-side and
      // the caller orders the stream; a second call on the same index may not start before the first returns - one mutex per index)
      HIP_TRY(hipStreamSynchronize(stream_));
      memcpy(h_q_.p, queries, qb);
      HIP_TRY(hipMemcpyAsync(q_buf_.p, h_q_.p, qb, hipMemcpyHostToDevice, stream_));
    } else {
      HIP_TRY(hipMemcpyAsync(q_buf_.p, queries, qb, hipMemcpyHostToDevice, stream_));
    }
    dq = q_buf_.as<float>();
  }

  // mode selection of VecSearchExecutor::Search (vec_search_executor.cpp:855-935)
  int mode = p.mode;
  int64_t limit = k;
  bool cap_local = false;
  if (mode == EPS_MODE_REFERENCE) {
    if (p.prefilter) {
      mode = EPS_MODE_FLAT;
    } else if (n_indexed_ < 512) {  // BruteforceThreshold, vec_search_executor.hpp:28
      mode = EPS_MODE_FLAT;
      cap_local = walk_limit == 0;  // result_size = min(size, limit, L_local_)  (:864); a candidate walk is cut by its caller
    } else {
      mode = EPS_MODE_GRAPH;
    }
  }
  if (mode == EPS_MODE_GRAPH && n_indexed_ <= 0) return fail(EPS_USER_ERROR, "search: graph mode requested but no graph is set");

  if (!run_buf_.reserve((size_t)nq * k * sizeof(u64))) return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (results)");
  u64* run_keys = run_buf_.as<u64>();
  HIP_TRY(hipEventRecord(ev0_, stream_));

  // results out (decided before the engines run: the matrix engine launches the result conversion itself, in front of its final
  // host sync, so that the device does not idle through that round trip)
  int64_t* d_ids = ids;
  float* d_dist = dist;
  int32_t* d_cnt = counts;
  const size_t ids_bytes = (size_t)nq * k * sizeof(int64_t), dist_bytes = (size_t)nq * k * sizeof(float), out_bytes = ids_bytes + dist_bytes + (size_t)nq * sizeof(int32_t);
  if (!out_dev) {
    if (!out_buf_.reserve(out_bytes)) return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory (outputs)");
    d_ids = out_buf_.as<int64_t>();
    d_dist = reinterpret_cast<float*>(out_buf_.as<char>() + ids_bytes);
    d_cnt = reinterpret_cast<int32_t*>(out_buf_.as<char>() + ids_bytes + dist_bytes);
  }
  result_finalized_ = false;
  auto finalize = [&]() {
    launch_finalize(run_keys, nq, k, id_base_, id_stride_, d_ids, d_dist, d_cnt, stream_);
    (void)hipEventRecord(ev1_, stream_);
    result_finalized_ = true;
  };

  int keff = k;
  if (mode == EPS_MODE_FLAT) {
    if (cap_local && p.local_queue < keff) keff = (int)p.local_queue;
    if (keff < k) launch_fill_u64(run_keys, nq * k, KEY_EMPTY, stream_);
    int engine = p.flat_engine;
    if (engine < EPS_FLAT_AUTO || engine > EPS_FLAT_MFMA_I8) return fail(EPS_USER_ERROR, "search: unknown flat engine");
    int bits = engine == EPS_FLAT_MFMA ? 16 : (engine == EPS_FLAT_MFMA_I8 ? 8 : 0);   // AUTO: the library picks the operand width too
    if (engine == EPS_FLAT_MFMA_I8) engine = EPS_FLAT_MFMA;
    if (engine == EPS_FLAT_AUTO) engine = flat_mfma_profitable(*this, nq, keff) ? EPS_FLAT_MFMA : EPS_FLAT_STREAM;
    // a filter on @distance needs exact distances wherever it is evaluated; the MFMA engine selects its seeds on
    // approximate keys, so such searches stay on the exact stream engine
    if (prog_len_ > 0 && prog_uses_dist_ && !prefilter_call_) engine = EPS_FLAT_STREAM;
    if (keff > 1024) engine = EPS_FLAT_STREAM;   // result pages (see flat_stream)
    int32_t rc;
    if (keff == k) {
      if (engine == EPS_FLAT_MFMA) {
        // (called by the engine in front of its final sync; again after a fall-back pass.  The callable captures locals of this
        // frame: the guard clears it on every way out, an exception from the engine included)
        struct PreSyncGuard {
          Index& ix;
          ~PreSyncGuard() {
            ix.pre_sync_ = nullptr;
            ix.pre_sync_nq_ = -1;
            ix.fin_ids_ = nullptr;
            ix.fin_dist_ = nullptr;
            ix.fin_cnt_ = nullptr;
          }
        } guard{*this};
        pre_sync_ = finalize;
        pre_sync_nq_ = nq;
        fin_ids_ = d_ids;
        fin_dist_ = d_dist;
        fin_cnt_ = d_cnt;
        rc = flat_mfma_search(*this, dq, nq, k, run_keys, false, bits);
      } else {
        rc = flat_stream(dq, nq, k, 0, n_rows_, run_keys, false);
      }
    } else {
      // narrower result (L_local cap): compute into a k_eff-wide list, then widen
      if (!tmp_buf_.reserve((size_t)nq * keff * sizeof(u64))) return fail(EPS_INFRA_UNEXPECTED_ERROR, "search: out of device memory");
      u64* narrow = tmp_buf_.as<u64>();
      rc = engine == EPS_FLAT_MFMA ? flat_mfma_search(*this, dq, nq, keff, narrow, false, bits)
                                   : flat_stream(dq, nq, keff, 0, n_rows_, narrow, false);
      if (rc == EPS_OK)
        HIP_TRY(hipMemcpy2DAsync(run_keys, (size_t)k * sizeof(u64), narrow, (size_t)keff * sizeof(u64),
                                 (size_t)keff * sizeof(u64), (size_t)nq, hipMemcpyDeviceToDevice, stream_));
    }
    if (rc != EPS_OK) return rc;
  } else {
    int64_t evals = 0;
    int32_t rc = graph_search(*this, dq, nq, k, p, run_keys, &evals, walk_limit);
    if (rc != EPS_OK) return rc;
  }
  (void)limit;

  if (!result_finalized_) finalize();
  if (!out_dev) {
    if (!(tune_int("EPS_HOST_STAGING", 1) == 0) && h_out_.reserve(out_bytes)) {   // one copy into page-locked memory, split on the host
      HIP_TRY(hipMemcpyAsync(h_out_.p, d_ids, out_bytes, hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipStreamSynchronize(stream_));
      const char* h = static_cast<const char*>(h_out_.p);
      memcpy(ids, h, ids_bytes);
      memcpy(dist, h + ids_bytes, dist_bytes);
      if (counts) memcpy(counts, h + ids_bytes + dist_bytes, (size_t)nq * sizeof(int32_t));
    } else {   // (no page-locked memory to be had: the pageable copies)
      HIP_TRY(hipMemcpyAsync(ids, d_ids, ids_bytes, hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipMemcpyAsync(dist, d_dist, dist_bytes, hipMemcpyDeviceToHost, stream_));
      if (counts) HIP_TRY(hipMemcpyAsync(counts, d_cnt, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
      HIP_TRY(hipStreamSynchronize(stream_));
    }
  }
  HIP_TRY(hipGetLastError());
  kring_valid_[kring_seq_ % KRING] = stats_.main_kernel_launches > 0;
  return EPS_OK;
}

int32_t Index::last_stats(eps_search_stats* out) {
  eps_search_stats s = stats_;
  // event timings are read lazily: the caller may have left the work in flight
  if (ev0_ && hipEventSynchronize(ev1_) == hipSuccess) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) s.kernel_ms = ms;
    if (s.main_kernel_launches > 0 && hipEventElapsedTime(&ms, evk0_, evk1_) == hipSuccess) s.main_kernel_ms = ms;
    double all = 0.0;
    for (int i = 0; i < stage_n_; ++i)
      if (hipEventElapsedTime(&ms, stage_ev_[i][0], stage_ev_[i][1]) == hipSuccess) all += ms;
    s.filter_ms_all = all;
  }
  (void)hipGetLastError();
  *out = s;
  return EPS_OK;
}

// main-kernel milliseconds of the most recent search calls (oldest first); synchronises the stream
int Index::kernel_times(double* ms_out, int cap) {
  if (!ms_out || cap <= 0) return 0;
  if (hipSetDevice(device_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int n = 0;
  const int64_t first = std::max<int64_t>(1, kring_seq_ - std::min<int64_t>(cap, KRING) + 1);
  for (int64_t q = first; q <= kring_seq_; ++q) {
    const int slot = (int)(q % KRING);
    if (!kring_valid_[slot]) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, kring_[slot][0], kring_[slot][1]) == hipSuccess) ms_out[n++] = ms;
    else (void)hipGetLastError();
  }
  return n;
}

}  // namespace eps

// ================================================================================================ C ABI
using eps::Index;
using eps::IndexBase;

extern "C" {

void eps_default_search_params(eps_search_params* p) {
  if (!p) return;
  p->mode = EPS_MODE_REFERENCE;
  p->flat_engine = EPS_FLAT_AUTO;
  p->prefilter = 0;         // Config::PreFilter{false}
  p->intra_threads = 4;     // Config::IntraQueryThreads{4}      (config/config.hpp:18)
  p->master_queue = 500;    // Config::MasterQueueSize{500}      (:19)
  p->local_queue = 500;     // Config::LocalQueueSize{500}       (:20)
  p->sync_interval = 15;    // Config::GlobalSyncInterval{15}    (:21)
  p->filter_in_traversal = 0;
  p->reserved = 0;
}

void eps_default_build_params(eps_build_params* p) {
  if (!p) return;
  p->search_length = 45;  // NSGConfig(45, 50, 300, 100), db/ann_graph_segment.cpp:29
  p->out_degree = 50;
  p->candidate_pool_size = 300;
  p->knng = 100;
  p->seed = 100;  // nsg.cpp:19
  p->reserved = 0;
}

// No C++ exception crosses the C ABI: allocation failures and anything else thrown below map to the reference's
// status codes (utils/error.hpp:11-41) with the text in eps_index_last_error.
static int32_t map_exception(IndexBase* ix) {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    return ix ? ix->fail(EPS_INFRA_UNEXPECTED_ERROR, "out of host memory") : EPS_INFRA_UNEXPECTED_ERROR;
  } catch (const std::exception& e) {
    return ix ? ix->fail(EPS_DB_UNEXPECTED_ERROR, std::string("unexpected: ") + e.what()) : EPS_DB_UNEXPECTED_ERROR;
  } catch (...) {
    return ix ? ix->fail(EPS_DB_UNEXPECTED_ERROR, "unexpected exception") : EPS_DB_UNEXPECTED_ERROR;
  }
}
#define IX(h) reinterpret_cast<IndexBase*>(h)
#define CIX(h) reinterpret_cast<const IndexBase*>(h)
#define GUARD(h, expr)             \
  do {                             \
    if (!(h)) return EPS_USER_ERROR; \
    try {                          \
      return (expr);               \
    } catch (...) {                \
      return map_exception(IX(h)); \
    }                              \
  } while (0)

int32_t eps_index_create(int64_t dim, int32_t metric, int32_t device, eps_index** out) {
  if (!out) return EPS_USER_ERROR;
  *out = nullptr;
  if (dim <= 0 || dim > 8192 || metric < 0 || metric > 2) return EPS_USER_ERROR;  // one query must fit in LDS next to the queues
  try {
    Index* ix = new Index(dim, metric, device);
    const int32_t rc = ix->init();
    if (rc != EPS_OK) {
      std::fprintf(stderr, "eps_index_create: %s\n", ix->last_error());
      delete ix;
      return rc;
    }
    *out = reinterpret_cast<eps_index*>(static_cast<IndexBase*>(ix));
    return EPS_OK;
  } catch (...) {
    return map_exception(nullptr);
  }
}
int32_t eps_index_create_sharded(int64_t dim, int32_t metric, const int32_t* devices, int32_t shards, eps_index** out) {
  if (!out) return EPS_USER_ERROR;
  *out = nullptr;
  if (dim <= 0 || dim > 8192 || metric < 0 || metric > 2 || !devices || shards <= 0 || shards > 16) return EPS_USER_ERROR;
  try {
    std::string err;
    IndexBase* g = eps::make_shard_group(dim, metric, devices, shards, &err);
    if (!g) {
      std::fprintf(stderr, "eps_index_create_sharded: %s\n", err.c_str());
      return EPS_INFRA_UNEXPECTED_ERROR;
    }
    *out = reinterpret_cast<eps_index*>(g);
    return EPS_OK;
  } catch (...) {
    return map_exception(nullptr);
  }
}
int32_t eps_index_destroy(eps_index* h) {
  try {
    delete reinterpret_cast<IndexBase*>(h);
    return EPS_OK;
  } catch (...) {
    return map_exception(nullptr);
  }
}
const char* eps_index_last_error(const eps_index* h) { return h ? CIX(h)->last_error() : "null handle"; }
int32_t eps_index_last_error_class(const eps_index* h) { return h ? CIX(h)->last_error_class() : EPS_ERRCLASS_OTHER; }
int32_t eps_index_set_stream(eps_index* h, void* s) { GUARD(h, IX(h)->set_stream(s)); }
int32_t eps_index_synchronize(eps_index* h) { GUARD(h, IX(h)->synchronize()); }
int32_t eps_index_attach_rows(eps_index* h, const float* rows, int64_t n) { GUARD(h, IX(h)->attach_rows(rows, n)); }
int32_t eps_index_append_rows(eps_index* h, const float* rows, int64_t n) { GUARD(h, IX(h)->append_rows(rows, n)); }
int32_t eps_index_attach_shard_rows(eps_index* h, int32_t shard, const float* rows, int64_t n_local) { GUARD(h, IX(h)->attach_shard_rows(shard, rows, n_local)); }
int32_t eps_index_clone_rows(eps_index* dst, eps_index* src, int64_t n) {
  if (!src) return EPS_USER_ERROR;
  GUARD(dst, IX(dst)->clone_rows(*IX(src), n));
}
int64_t eps_index_row_count(const eps_index* h) { return h ? CIX(h)->row_count() : -1; }
int32_t eps_index_set_id_map(eps_index* h, int64_t b, int64_t s) { GUARD(h, IX(h)->set_id_map(b, s)); }
int32_t eps_index_set_deleted(eps_index* h, const uint8_t* bits, int64_t nbytes) { GUARD(h, IX(h)->set_deleted(bits, nbytes)); }
int32_t eps_index_set_int_filter(eps_index* h, const void* col, int64_t stride, int32_t width, int32_t op, int64_t c) {
  GUARD(h, IX(h)->set_int_filter(col, stride, width, op, c));
}
int32_t eps_index_set_filter_program(eps_index* h, const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride, int64_t n_rows) {
  GUARD(h, IX(h)->set_filter_program(ops, nops, rows, stride, n_rows, 0));
}
int32_t eps_index_set_filter_program_ex(eps_index* h, const eps_filter_op* ops, int32_t nops, const void* rows, int64_t stride, int64_t n_rows,
                                        int32_t flags) {
  GUARD(h, IX(h)->set_filter_program(ops, nops, rows, stride, n_rows, flags));
}
int32_t eps_index_search_walk(eps_index* h, const float* q, int64_t nq, int32_t limit, int32_t cap, const eps_search_params* p, int64_t* ids,
                              float* dist, int32_t* counts) {
  if (limit <= 0 || cap < limit) return EPS_USER_ERROR;
  GUARD(h, IX(h)->search(q, nq, cap, p, ids, dist, counts, limit));
}
int32_t eps_index_select_edges(eps_index* h, const int64_t* nodes, int64_t m, const int64_t* cands, int32_t cands_per_node, int32_t depth,
                               int32_t out_degree, int64_t* out_ids, int32_t* out_deg) {
  if (!h) return EPS_USER_ERROR;
  Index* ix = dynamic_cast<Index*>(IX(h));
  if (!ix) return IX(h)->fail(EPS_DB_UNSUPPORTED_ERROR, "select_edges: single-device indices only");
  try {
    if (hipSetDevice(ix->device_) != hipSuccess) return ix->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    return eps::select_edges(*ix, nodes, m, cands, cands_per_node, depth, out_degree, out_ids, out_deg);
  } catch (...) {
    return map_exception(ix);
  }
}
int32_t eps_index_inter_insert(eps_index* h, const int64_t* ids, const int32_t* deg, int64_t n, int32_t out_degree, int64_t* out_ids,
                               int32_t* out_deg) {
  if (!h) return EPS_USER_ERROR;
  Index* ix = dynamic_cast<Index*>(IX(h));
  if (!ix) return IX(h)->fail(EPS_DB_UNSUPPORTED_ERROR, "inter_insert: single-device indices only");
  try {
    if (hipSetDevice(ix->device_) != hipSuccess) return ix->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    return eps::inter_insert(*ix, ids, deg, n, out_degree, out_ids, out_deg);
  } catch (...) {
    return map_exception(ix);
  }
}
int32_t eps_index_knn_graph(eps_index* h, int64_t n, const eps_build_params* p, int64_t* out_ids) {
  if (!h || !out_ids) return EPS_USER_ERROR;
  Index* ix = dynamic_cast<Index*>(IX(h));
  if (!ix) return IX(h)->fail(EPS_DB_UNSUPPORTED_ERROR, "knn_graph: single-device indices only");
  try {
    eps_build_params bp;
    if (p) bp = *p; else eps_default_build_params(&bp);
    if (n < 2 || n > ix->row_count()) return ix->fail(EPS_USER_ERROR, "knn_graph: n must be in [2, rows]");
    if (hipSetDevice(ix->device_) != hipSuccess) return ix->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    eps::BuildStage st;
    st.stop_after = 1;
    st.out_ids = out_ids;
    return eps::graph_build(*ix, n, bp, &st);
  } catch (...) {
    return map_exception(ix);
  }
}
int32_t eps_index_link(eps_index* h, int64_t n, const int64_t* knn, int64_t navigation_point, const eps_build_params* p, int64_t* out_ids,
                       int32_t* out_deg, int64_t* nav_out) {
  if (!h || !out_ids || !out_deg) return EPS_USER_ERROR;
  Index* ix = dynamic_cast<Index*>(IX(h));
  if (!ix) return IX(h)->fail(EPS_DB_UNSUPPORTED_ERROR, "link: single-device indices only");
  try {
    eps_build_params bp;
    if (p) bp = *p; else eps_default_build_params(&bp);
    if (n < 2 || n > ix->row_count()) return ix->fail(EPS_USER_ERROR, "link: n must be in [2, rows]");
    if (hipSetDevice(ix->device_) != hipSuccess) return ix->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    eps::BuildStage st;
    st.knn_in = knn;
    st.nav_in = navigation_point;
    st.stop_after = 2;
    st.out_ids = out_ids;
    st.out_deg = out_deg;
    st.nav_out = nav_out;
    return eps::graph_build(*ix, n, bp, &st);
  } catch (...) {
    return map_exception(ix);
  }
}
int32_t eps_index_load_table(eps_index* h, const char* path, const eps_table_layout* layout, int64_t* n_out) {
  if (!h) return EPS_USER_ERROR;
  Index* ix = dynamic_cast<Index*>(IX(h));
  if (!ix) return IX(h)->fail(EPS_DB_UNSUPPORTED_ERROR, "load_table: single-device indices only");
  try {
    if (hipSetDevice(ix->device_) != hipSuccess) return ix->fail(EPS_INFRA_UNEXPECTED_ERROR, "hipSetDevice");
    return ix->load_table(path, layout, n_out);
  } catch (...) {
    return map_exception(ix);
  }
}
int32_t eps_index_build(eps_index* h, int64_t n, const eps_build_params* p) { GUARD(h, IX(h)->build(n, p)); }
int32_t eps_index_set_graph(eps_index* h, int64_t n, const int64_t* off, const int64_t* nbr, int64_t nav) {
  GUARD(h, IX(h)->set_graph(n, off, nbr, nav));
}
int32_t eps_index_graph_info(const eps_index* h, int64_t* n, int64_t* e, int64_t* nav) { return h ? CIX(h)->graph_info(n, e, nav) : EPS_USER_ERROR; }
int32_t eps_index_get_graph(const eps_index* h, int64_t* off, int64_t* nbr) { return h ? CIX(h)->get_graph(off, nbr) : EPS_USER_ERROR; }
int32_t eps_index_save_graph(eps_index* h, const char* path) { GUARD(h, IX(h)->save_graph(path)); }
int32_t eps_index_load_graph(eps_index* h, const char* path) { GUARD(h, IX(h)->load_graph(path)); }
int32_t eps_index_search(eps_index* h, const float* q, int64_t nq, int32_t k, const eps_search_params* p, int64_t* ids,
                         float* dist, int32_t* counts) {
  GUARD(h, IX(h)->search(q, nq, k, p, ids, dist, counts));
}
int32_t eps_index_last_stats(const eps_index* h, eps_search_stats* out) {
  if (!h || !out) return EPS_USER_ERROR;
  return const_cast<IndexBase*>(CIX(h))->last_stats(out);
}
int32_t eps_index_kernel_times(eps_index* h, double* ms_out, int32_t cap) { return h ? IX(h)->kernel_times(ms_out, cap) : 0; }

int32_t eps_normalize_rows(float* rows, int64_t n, int64_t dim, int32_t only_if_nonzero, int32_t device, void* stream) {
  if (n < 0 || dim <= 0 || (n > 0 && !rows)) return EPS_USER_ERROR;
  if (n == 0) return EPS_OK;
  if (hipSetDevice(device) != hipSuccess) return EPS_INFRA_UNEXPECTED_ERROR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps::is_device_ptr(rows)) {
    eps::launch_normalize(rows, n, (int)dim, only_if_nonzero != 0, s);
    return hipGetLastError() == hipSuccess ? EPS_OK : EPS_INFRA_UNEXPECTED_ERROR;
  }
  float* d = nullptr;
  const size_t bytes = (size_t)n * dim * sizeof(float);
  if (hipMalloc(&d, bytes) != hipSuccess) return EPS_INFRA_UNEXPECTED_ERROR;
  bool ok = hipMemcpyAsync(d, rows, bytes, hipMemcpyHostToDevice, s) == hipSuccess;
  if (ok) eps::launch_normalize(d, n, (int)dim, only_if_nonzero != 0, s);
  ok = ok && hipMemcpyAsync(rows, d, bytes, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  (void)hipFree(d);
  return ok ? EPS_OK : EPS_INFRA_UNEXPECTED_ERROR;
}

int32_t eps_merge_topk(const float* dist, const int64_t* ids, int32_t shards, int64_t nq, int32_t k, float* out_dist,
                       int64_t* out_ids, int32_t device, void* stream) {
  if (!dist || !ids || !out_dist || !out_ids || shards <= 0 || shards > 16 || nq < 0 || k <= 0) return EPS_USER_ERROR;
  if (nq == 0) return EPS_OK;
  if (hipSetDevice(device) != hipSuccess) return EPS_INFRA_UNEXPECTED_ERROR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool dev = eps::is_device_ptr(dist);
  if (dev != eps::is_device_ptr(ids) || dev != eps::is_device_ptr(out_dist) || dev != eps::is_device_ptr(out_ids)) return EPS_USER_ERROR;
  if (dev) {
    eps::launch_merge_shards(dist, ids, shards, nq, k, out_dist, out_ids, s);
    return hipGetLastError() == hipSuccess ? EPS_OK : EPS_INFRA_UNEXPECTED_ERROR;
  }
  const size_t in_n = (size_t)shards * nq * k, out_n = (size_t)nq * k;
  char* d = nullptr;
  if (hipMalloc(&d, in_n * 12 + out_n * 12 + 64) != hipSuccess) return EPS_INFRA_UNEXPECTED_ERROR;
  int64_t* d_ids = reinterpret_cast<int64_t*>(d);
  int64_t* d_oids = d_ids + in_n;
  float* d_dist = reinterpret_cast<float*>(d_oids + out_n);
  float* d_odist = d_dist + in_n;
  bool ok = hipMemcpyAsync(d_ids, ids, in_n * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
            hipMemcpyAsync(d_dist, dist, in_n * 4, hipMemcpyHostToDevice, s) == hipSuccess;
  if (ok) eps::launch_merge_shards(d_dist, d_ids, shards, nq, k, d_odist, d_oids, s);
  ok = ok && hipMemcpyAsync(out_ids, d_oids, out_n * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
       hipMemcpyAsync(out_dist, d_odist, out_n * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
       hipStreamSynchronize(s) == hipSuccess;
  (void)hipFree(d);
  return ok ? EPS_OK : EPS_INFRA_UNEXPECTED_ERROR;
}

// the same merge over ONE gathered buffer: shard s contributed `shard_stride_bytes` bytes holding int64 ids[nq][k] at
// offset 0 and float dist[nq][k] at `dist_offset_bytes` (what a single all-gather of a packed per-rank buffer delivers)
int32_t eps_merge_topk_packed(const void* gathered, int64_t shard_stride_bytes, int64_t dist_offset_bytes, int32_t shards, int64_t nq,
                              int32_t k, float* out_dist, int64_t* out_ids, int32_t device, void* stream) {
  if (!gathered || !out_dist || !out_ids || shards <= 0 || shards > 16 || nq < 0 || k <= 0) return EPS_USER_ERROR;
  if (shard_stride_bytes < dist_offset_bytes + nq * k * 4 || dist_offset_bytes < nq * k * 8 || (dist_offset_bytes & 3) || (shard_stride_bytes & 7))
    return EPS_USER_ERROR;
  if (nq == 0) return EPS_OK;
  if (!eps::is_device_ptr(gathered) || !eps::is_device_ptr(out_dist) || !eps::is_device_ptr(out_ids)) return EPS_USER_ERROR;
  if (hipSetDevice(device) != hipSuccess) return EPS_INFRA_UNEXPECTED_ERROR;
  const char* base = static_cast<const char*>(gathered);
  eps::launch_merge_shards(reinterpret_cast<const float*>(base + dist_offset_bytes), reinterpret_cast<const int64_t*>(base), shards, nq, k,
                           out_dist, out_ids, static_cast<hipStream_t>(stream), shard_stride_bytes);
  return hipGetLastError() == hipSuccess ? EPS_OK : EPS_INFRA_UNEXPECTED_ERROR;
}

int32_t eps_set_tuning(const char* name, const char* value) {
  eps::tune_set(name, value);
  return EPS_OK;
}

}  // extern "C"
