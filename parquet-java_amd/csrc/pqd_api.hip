// C ABI of libpqdecode_hip.so (see include/pqd.h): argument checks, dispatch to the kernels, and
// the host-memory convenience paths (stage -> decode -> copy back).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pqd.h"
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"

// Scratch of pqd_decode_pages_dev_noalloc: a chunk pool, one plan record per page and the counter
// block of the planned hybrid / dictionary paths, in a file-scope device array (every GPU the
// library runs on holds its own copy until the process ends).  Host code never uses the array's
// name as an address: the per-device address comes from hipGetSymbolAddress (noalloc_scratch).
constexpr uint32_t NOALLOC_MAX_PAGES = 2048;
constexpr uint32_t NOALLOC_POOL_CHUNKS = 8192;  // chunk p for page p, the rest shared
constexpr size_t NOALLOC_POOL_BYTES = (size_t)NOALLOC_POOL_CHUNKS * pqd::SEG_CHUNK * sizeof(pqd::RleSeg);
constexpr size_t NOALLOC_PLANS_BYTES = (size_t)NOALLOC_MAX_PAGES * sizeof(pqd::RlePagePlan);
constexpr size_t NOALLOC_COUNTER_BYTES = pqd::PLAN_COUNTER_BYTES + (size_t)pqd::DP_MAX_SLOTS * 12;  // counters, plane key table, plane slot list
constexpr uint32_t NOALLOC_PLANE_SLOTS = 8;  // plane copies of 8-byte dictionaries swept through LDS
constexpr size_t NOALLOC_PLANE_BYTES = (size_t)NOALLOC_PLANE_SLOTS * pqd::DP_SLOT_BYTES;
constexpr size_t NOALLOC_BYTES = NOALLOC_POOL_BYTES + NOALLOC_PLANS_BYTES + NOALLOC_COUNTER_BYTES + NOALLOC_PLANE_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

namespace {

thread_local std::string g_last_error;

int fail(int status, const char* what, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess)
    snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
  else
    snprintf(buf, sizeof buf, "%s", what);
  g_last_error = buf;
  return status;
}

#define PQD_HIP(expr)                                            \
  do {                                                           \
    hipError_t e_ = (expr);                                      \
    if (e_ != hipSuccess) return fail(PQD_HIP_ERROR, #expr, e_); \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Growable device buffer owned by one host thread (used only by the host-memory entry points).
struct Scratch {
  void* p = nullptr;
  size_t cap = 0;
  int device = -1;
  int ensure(size_t bytes) {
    int dev = 0;
    PQD_HIP(hipGetDevice(&dev));
    if (p && device == dev && cap >= bytes) return PQD_OK;
    if (p) {
      (void)hipFree(p);
      p = nullptr;
      cap = 0;
    }
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
    PQD_HIP(hipMalloc(&p, want));
    cap = want;
    device = dev;
    return PQD_OK;
  }
};

// Everything the host-memory entry points keep per calling thread: device staging buffers and a
// stream of the thread's own (created on first use, on the device current then).  Calls from
// different host threads therefore never share a stream, staging memory or plan scratch.
struct HostCtx {
  Scratch in, out, dict, descs, results;
  hipStream_t stream = nullptr;
  int stream_device = -1;
  int ensure_stream(hipStream_t* s);
  void release();
  ~HostCtx() { release(); }
};
thread_local HostCtx g_ctx;

int check_device_available() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(PQD_NO_DEVICE, "no HIP device visible");
  return PQD_OK;
}

// Device scratch of the planned RLE paths: one run-plan record per page and the chunk pool.
// One per (device, stream).  The kernels of one call (counter reset, plan, expansion / gather,
// left-over pages) must reach the stream as one uninterrupted sequence: stream order then makes the
// next call's reset wait for this call's last kernel.  `mu` is held by the calling thread from the
// lookup (and a possible growth, which first drains the stream) until the last launch is enqueued,
// so host threads that share a stream (the NULL stream included) queue up here instead of
// interleaving their launches or freeing buffers another thread is about to use.
// Grown on demand, never shrunk, released with the stream or the process.
struct PlanScratch {
  std::mutex mu;
  void* plans = nullptr;
  uint32_t cap_pages = 0;
  void* gen_pool = nullptr;    // chunk pool: chunk p for page p, the rest handed out through counters[0]
  uint32_t gen_chunks = 0;
  uint32_t* counters = nullptr;  // TWO counter blocks (pqd_rle_plan.hpp: counters, then the key table and the list of the
                                 // dictionary plane slots).  A call uses block `which`, which is zero when the call
                                 // begins, and its last launch zeroes the other one for the call after it, so no call
                                 // begins with a memset of its own (a launch of 5 us in front of a 17 us kernel).
  int which = 0;
  bool clean[2] = {false, false};  // the block was zeroed after its last use
  bool captured = false;           // a call on this stream was captured into a graph: replays run outside this
                                   // book-keeping, so from then on every call zeroes its block itself
  void* plane_area = nullptr;    // plane copies of 8-byte dictionaries, plane_slots x DP_SLOT_BYTES
  uint32_t plane_slots = 0;
};
constexpr size_t COUNTER_BYTES = pqd::PLAN_COUNTER_BYTES;
constexpr size_t PLANE_KEYS_BYTES = (size_t)pqd::DP_MAX_SLOTS * 8, PLANE_LIST_BYTES = (size_t)pqd::DP_MAX_SLOTS * 4;
constexpr size_t COUNTER_BLOCK_BYTES = COUNTER_BYTES + PLANE_KEYS_BYTES + PLANE_LIST_BYTES;

pqd::DictPlanes planes_of(uint32_t* counters, void* area, uint32_t slots) {
  pqd::DictPlanes dp;
  dp.keys = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(counters) + COUNTER_BYTES);
  dp.list = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(counters) + COUNTER_BYTES + PLANE_KEYS_BYTES);
  dp.area = static_cast<uint8_t*>(area);
  dp.n_slots = slots;
  return dp;
}
std::mutex g_plan_mutex;  // guards the map itself (nodes of a std::map do not move)
std::map<std::pair<int, void*>, PlanScratch> g_plan_scratch;

struct PlanLease {
  PlanScratch* ps = nullptr;
  std::unique_lock<std::mutex> lock;
};

// Plans for n_pages pages and a chunk pool of three chunks per page plus a shared reserve (a
// 20000-value level page needs 1-6 chunks, a 1 M-value page about 31; a page that finds the pool
// empty is decoded by the generic kernel).  Returns with the scratch locked.
int plan_scratch_acquire(hipStream_t stream, uint32_t n_pages, uint32_t want_plane_slots, PlanLease* lease) {
  int dev = 0;
  PQD_HIP(hipGetDevice(&dev));
  PlanScratch* ps;
  {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    ps = &g_plan_scratch[std::make_pair(dev, (void*)stream)];
  }
  std::unique_lock<std::mutex> lock(ps->mu);
  const uint64_t want = (uint64_t)n_pages * 3 + 32768;  // (dictionary pages split the tail behind the page slots into two halves)
  if (want_plane_slots > pqd::DP_MAX_SLOTS) want_plane_slots = pqd::DP_MAX_SLOTS;
  if (ps->cap_pages < n_pages || ps->gen_chunks < want || !ps->counters || ps->plane_slots < want_plane_slots) {
    // wait for work that may still use the old buffers, then replace them
    PQD_HIP(hipStreamSynchronize(stream));
    if (ps->cap_pages < n_pages) {
      if (ps->plans) (void)hipFree(ps->plans);
      ps->plans = nullptr;
      ps->cap_pages = 0;
      const uint32_t pages = n_pages + n_pages / 4 + 64;
      PQD_HIP(hipMalloc(&ps->plans, (size_t)pages * sizeof(pqd::RlePagePlan)));
      ps->cap_pages = pages;
    }
    if (ps->gen_chunks < want) {
      if (ps->gen_pool) (void)hipFree(ps->gen_pool);
      ps->gen_pool = nullptr;
      ps->gen_chunks = 0;
      const uint64_t chunks = want + want / 4;
      PQD_HIP(hipMalloc(&ps->gen_pool, (size_t)chunks * pqd::SEG_CHUNK * sizeof(pqd::RleSeg)));
      ps->gen_chunks = (uint32_t)(chunks > 0xFFFFFFFFull ? 0xFFFFFFFFull : chunks);
    }
    if (!ps->counters) {
      PQD_HIP(hipMalloc((void**)&ps->counters, 2 * COUNTER_BLOCK_BYTES));
      ps->clean[0] = ps->clean[1] = false;
    }
    if (ps->plane_slots < want_plane_slots) {
      if (ps->plane_area) (void)hipFree(ps->plane_area);
      ps->plane_area = nullptr;
      ps->plane_slots = 0;
      PQD_HIP(hipMalloc(&ps->plane_area, (size_t)want_plane_slots * pqd::DP_SLOT_BYTES));
      ps->plane_slots = want_plane_slots;
    }
  }
  lease->ps = ps;
  lease->lock = std::move(lock);
  return PQD_OK;
}

// The counter block of this call (zeroed here only if the call before did not get to do it) ...
int counters_begin(PlanScratch& ps, hipStream_t s, uint32_t** counters, uint32_t** other) {
  // A sequence captured into a graph is replayed any number of times, in an order the host state here knows
  // nothing of: a captured call zeroes its block inside the sequence (as does every call after it).
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  PQD_HIP(hipStreamIsCapturing(s, &cap));
  if (cap != hipStreamCaptureStatusNone) ps.captured = true;
  uint32_t* base = ps.counters + (size_t)ps.which * (COUNTER_BLOCK_BYTES / 4);
  *other = ps.counters + (size_t)(ps.which ^ 1) * (COUNTER_BLOCK_BYTES / 4);
  if (!ps.clean[ps.which] || ps.captured) PQD_HIP(hipMemsetAsync(base, 0, COUNTER_BYTES + PLANE_KEYS_BYTES, s));
  ps.clean[ps.which] = false;  // in use from now on
  *counters = base;
  return PQD_OK;
}
// ... and, once the call's last launch (which zeroes the other block) is enqueued, the switch to that block.
void counters_end(PlanScratch& ps) {
  if (ps.captured) return;  // a replay uses the block it was captured with: no switching any more
  ps.clean[ps.which ^ 1] = true;
  ps.which ^= 1;
}
constexpr uint32_t COUNTER_ZERO_WORDS = (uint32_t)((COUNTER_BYTES + PLANE_KEYS_BYTES) / 4);

// Frees the scratch kept for `stream` on any device (the stream must be idle).
void plan_scratch_drop(void* stream) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  for (auto it = g_plan_scratch.begin(); it != g_plan_scratch.end();) {
    if (it->first.second == stream) {
      {
        std::lock_guard<std::mutex> inner(it->second.mu);
        if (it->second.plans) (void)hipFree(it->second.plans);
        if (it->second.gen_pool) (void)hipFree(it->second.gen_pool);
        if (it->second.counters) (void)hipFree(it->second.counters);
        if (it->second.plane_area) (void)hipFree(it->second.plane_area);
      }
      it = g_plan_scratch.erase(it);
    } else {
      ++it;
    }
  }
}

int HostCtx::ensure_stream(hipStream_t* s) {
  int dev = 0;
  PQD_HIP(hipGetDevice(&dev));
  if (stream && stream_device != dev) {  // the thread moved to another device: its stream goes
    (void)hipStreamSynchronize(stream);
    plan_scratch_drop(stream);
    (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
  if (!stream) {
    PQD_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    stream_device = dev;
  }
  *s = stream;
  return PQD_OK;
}

void HostCtx::release() {
  for (Scratch* b : {&in, &out, &dict, &descs, &results}) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
    b->device = -1;
  }
  if (stream) {
    (void)hipStreamSynchronize(stream);
    plan_scratch_drop(stream);
    (void)hipStreamDestroy(stream);
    stream = nullptr;
    stream_device = -1;
  }
}

// Device address of `scratch` on the current device (looked up once per device, with that device
// current: each GPU has its own copy of the array at its own address).
constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_noalloc_mutex;  // guards the tables; also held while a call's launches are enqueued
unsigned char* g_noalloc_ptr[NOALLOC_MAX_DEVICES];
// Calls on different streams of one GPU share that GPU's array, so they are ordered on the device: every call
// records this event behind its last launch and the next call's stream waits for it first (created with the
// address look-up, once per device; no allocation on the call path).
hipEvent_t g_noalloc_done[NOALLOC_MAX_DEVICES];
bool g_noalloc_recorded[NOALLOC_MAX_DEVICES];

int noalloc_scratch(unsigned char** out, int* dev_out = nullptr) {
  int dev = 0;
  PQD_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(PQD_BAD_ARG, "device index beyond the library's table");
  if (!g_noalloc_ptr[dev]) {
    void* p = nullptr;
    PQD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(PQD_HIP_ERROR, "hipGetSymbolAddress returned no address for the scratch array");
    PQD_HIP(hipEventCreateWithFlags(&g_noalloc_done[dev], hipEventDisableTiming));
    g_noalloc_recorded[dev] = false;
    g_noalloc_ptr[dev] = static_cast<unsigned char*>(p);
  }
  *out = g_noalloc_ptr[dev];
  if (dev_out) *dev_out = dev;
  return PQD_OK;
}

// Optional per-launch timing of the dictionary path (pqd_set_phase_timing): one set of four
// events per timed call, kept until pqd_phase_ms_mean reads them.
struct PhaseTimer {
  bool enabled = false;
  bool gather_only = false;  // record only the two events around the gather launch
  int device = -1;
  std::vector<hipEvent_t> events;  // 4 per call; the first `used` sets are recorded
  size_t used = 0;
};
// Several series side by side (pqd_set_phase_series): a benchmark that alternates calls of different kinds keeps one
// series per kind.
constexpr int PHASE_SERIES = 4;
thread_local PhaseTimer g_phase_series[PHASE_SERIES];
thread_local int g_phase_cur = 0;
#define g_phase (g_phase_series[g_phase_cur])
constexpr size_t PHASE_MAX_CALLS = 4096;

int phase_mark(int i, hipStream_t s) {
  PhaseTimer& t = g_phase;
  if (!t.enabled || t.used >= PHASE_MAX_CALLS) return PQD_OK;
  if (t.gather_only && i == 3) {  // events 0 and 3 of the set stay unrecorded
    ++t.used;
    return PQD_OK;
  }
  if (i == 0) {
    int dev = 0;
    PQD_HIP(hipGetDevice(&dev));
    if (t.device != dev) {
      for (auto e : t.events) (void)hipEventDestroy(e);
      t.events.clear();
      t.used = 0;
      t.device = dev;
    }
    while (t.events.size() < (t.used + 1) * 4) {
      hipEvent_t e;
      PQD_HIP(hipEventCreate(&e));
      t.events.push_back(e);
    }
  }
  if (t.gather_only) return PQD_OK;  // (mark 0 made sure the events exist; the gather launch records 1 and 2)
  PQD_HIP(hipEventRecord(t.events[t.used * 4 + i], s));
  if (i == 3) ++t.used;
  return PQD_OK;
}

// Mode 2 of the phase timing: the two events of the current set that the gather launch itself records (start and end
// of the kernel, attached to its dispatch: no event packets in front of or behind it).  Null when the mode is off.
void phase_gather_events(hipEvent_t* start, hipEvent_t* stop) {
  *start = *stop = nullptr;
  PhaseTimer& t = g_phase;
  if (!t.enabled || !t.gather_only || t.used >= PHASE_MAX_CALLS || t.events.size() < (t.used + 1) * 4) return;
  *start = t.events[t.used * 4 + 1];
  *stop = t.events[t.used * 4 + 2];
}

// Which kernels decode dictionary pages: the planned path (header-walk kernel + LDS-sliced gather,
// with the generic kernel for the pages the plan leaves to it) unless PQD_DICT_PATH=generic asks
// for the single generic kernel.  Read on every call so that a test can switch inside one process.
bool use_sliced_dictionary_path() {
  const char* v = getenv("PQD_DICT_PATH");
  return !(v && strcmp(v, "generic") == 0);
}

// Chunks of the pool the kernels may use: all of them, unless PQD_POOL_CHUNKS sets a lower bound
// (pages that find no room are decoded by the generic kernel; the tests use it to reach that path
// without gigabyte pages).
uint32_t usable_pool_chunks(uint32_t have) {
  const char* v = getenv("PQD_POOL_CHUNKS");
  if (!v) return have;
  const long lim = atol(v);
  return lim < 0 ? 0u : ((unsigned long)lim < have ? (uint32_t)lim : have);
}

// Plane slots a dictionary call of n_pages pages asks for, and how many of those the kernels may use
// (PQD_PLANE_SLOTS lowers or raises it; 0 = none: the tests reach the gather-from-memory path with it).
uint32_t wanted_plane_slots(uint32_t n_pages) {
  uint32_t n = n_pages / 16;
  n = n < 4 ? 4 : n;
  if (const char* v = getenv("PQD_PLANE_SLOTS")) {
    const long lim = atol(v);
    n = lim < 0 ? 0u : (uint32_t)lim;
  }
  return n > pqd::DP_MAX_SLOTS ? pqd::DP_MAX_SLOTS : n;
}
uint32_t usable_plane_slots(uint32_t have, uint32_t n_pages) {
  const uint32_t want = wanted_plane_slots(n_pages);
  return want < have ? want : have;
}

// The same switch for plain hybrid pages: PQD_RLE_PATH=generic selects the single generic kernel.
bool use_planned_rle_path() {
  const char* v = getenv("PQD_RLE_PATH");
  return !(v && strcmp(v, "generic") == 0);
}

// PQD_RLE_SMALL=0 keeps small hybrid pages on the planned kernels (k_rle_small_pages is skipped).
bool use_small_rle_kernel() {
  const char* v = getenv("PQD_RLE_SMALL");
  return !(v && strcmp(v, "0") == 0);
}

int class_of(const pqd_page_desc& d) {  // pages that can share a launch
  return ((int)d.encoding << 8) | (int)(d.encoding == PQD_ENC_BYTE_STREAM_SPLIT || d.encoding == PQD_ENC_PLAIN ? 0 : d.elem_size);
}

}  // namespace

extern "C" {

int pqd_version(void) { return PQD_VERSION; }

const char* pqd_status_string(int status) {
  switch (status) {
    case PQD_OK: return "ok";
    case PQD_EOF: return "EOFException: stream shorter than the decode needs";
    case PQD_BAD_ARG: return "IllegalArgumentException: argument outside the contract";
    case PQD_PAST_END: return "IllegalArgumentException: Reading past RLE/BitPacking stream.";
    case PQD_DECODE: return "ParquetDecodingException: malformed page";
    case PQD_DICT_OOB: return "ParquetDecodingException: dictionary id out of range";
    case PQD_HIP_ERROR: return "HIP runtime error";
    case PQD_NO_DEVICE: return "no usable gfx950 device";
    default: return "unknown status";
  }
}

const char* pqd_last_error(void) { return g_last_error.c_str(); }

int pqd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  int usable = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
    if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
  }
  return usable;
}

int pqd_init(int device) {
  int st = check_device_available();
  if (st) return st;
  hipDeviceProp_t prop;
  PQD_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(PQD_NO_DEVICE, "device is not gfx950");
  PQD_HIP(hipSetDevice(device));
  {  // the address of this device's copy of the no-allocation scratch
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    unsigned char* p;
    if ((st = noalloc_scratch(&p))) return st;
  }
  return PQD_OK;
}

int pqd_dev_alloc(void** p, size_t bytes) {
  if (!p) return fail(PQD_BAD_ARG, "pqd_dev_alloc: null out pointer");
  PQD_HIP(hipMalloc(p, bytes ? bytes : 1));
  return PQD_OK;
}
int pqd_dev_free(void* p) {
  if (p) PQD_HIP(hipFree(p));
  return PQD_OK;
}
int pqd_host_alloc_pinned(void** p, size_t bytes) {
  if (!p) return fail(PQD_BAD_ARG, "pqd_host_alloc_pinned: null out pointer");
  PQD_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
  return PQD_OK;
}
int pqd_host_free_pinned(void* p) {
  if (p) PQD_HIP(hipHostFree(p));
  return PQD_OK;
}
int pqd_alloc_pinned(void** p, size_t bytes) { return pqd_host_alloc_pinned(p, bytes); }
int pqd_free_pinned(void* p) { return pqd_host_free_pinned(p); }
int pqd_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes) PQD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  return PQD_OK;
}
int pqd_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes) PQD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  return PQD_OK;
}
int pqd_memset_dev(void* dst, int value, size_t bytes, void* stream) {
  if (bytes) PQD_HIP(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
  return PQD_OK;
}
int pqd_stream_create(void** stream) {
  if (!stream) return fail(PQD_BAD_ARG, "pqd_stream_create: null out pointer");
  hipStream_t s;
  PQD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return PQD_OK;
}
int pqd_stream_destroy(void* stream) {
  if (!stream) return PQD_OK;
  PQD_HIP(hipStreamSynchronize(as_stream(stream)));
  plan_scratch_drop(stream);  // the scratch the planned paths kept for this stream (on any device) goes with it
  PQD_HIP(hipStreamDestroy(as_stream(stream)));
  return PQD_OK;
}
int pqd_stream_sync(void* stream) {
  PQD_HIP(hipStreamSynchronize(as_stream(stream)));
  return PQD_OK;
}
int pqd_event_create(void** ev) {
  if (!ev) return fail(PQD_BAD_ARG, "pqd_event_create: null out pointer");
  hipEvent_t e;
  PQD_HIP(hipEventCreate(&e));
  *ev = e;
  return PQD_OK;
}
int pqd_event_destroy(void* ev) {
  if (ev) PQD_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
  return PQD_OK;
}
int pqd_event_record(void* ev, void* stream) {
  PQD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), as_stream(stream)));
  return PQD_OK;
}
int pqd_event_sync(void* ev) {
  PQD_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev)));
  return PQD_OK;
}
int pqd_set_phase_timing(int enable) {
  PhaseTimer& t = g_phase;
  if (enable != 0 && t.used != 0 && t.gather_only != (enable == 2))
    return fail(PQD_BAD_ARG, "pqd_set_phase_timing: read the running series (pqd_phase_ms_mean) before changing the mode");
  t.enabled = enable != 0;  // pausing keeps what has been recorded
  if (enable != 0) t.gather_only = enable == 2;
  return PQD_OK;
}

int pqd_set_phase_series(int series) {
  if (series < 0 || series >= PHASE_SERIES) return fail(PQD_BAD_ARG, "pqd_set_phase_series: series 0..3");
  g_phase_cur = series;
  return PQD_OK;
}

int pqd_phase_ms_mean(float ms[3], int* n_calls) {
  if (!ms) return fail(PQD_BAD_ARG, "null output");
  ms[0] = ms[1] = ms[2] = 0.f;
  PhaseTimer& t = g_phase;
  if (n_calls) *n_calls = (int)t.used;
  if (t.used == 0) return PQD_OK;
  double sum[3] = {0, 0, 0};
  for (size_t c = 0; c < t.used; ++c) {
    PQD_HIP(hipEventSynchronize(t.events[c * 4 + (t.gather_only ? 2 : 3)]));
    for (int i = 0; i < 3; ++i) {
      if (t.gather_only && i != 1) continue;
      float x = 0.f;
      PQD_HIP(hipEventElapsedTime(&x, t.events[c * 4 + i], t.events[c * 4 + i + 1]));
      sum[i] += x;
    }
  }
  for (int i = 0; i < 3; ++i) ms[i] = (float)(sum[i] / (double)t.used);
  t.used = 0;
  return PQD_OK;
}

int pqd_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return fail(PQD_BAD_ARG, "pqd_event_elapsed_ms: null out pointer");
  PQD_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return PQD_OK;
}

/* ---- bulk bit-unpack ----------------------------------------------------------------------- */

static int check_unpack_args(int bit_width, int max_width, const void* in, size_t in_len, const void* out,
                             size_t n_values, size_t* total) {
  if (bit_width < 0 || bit_width > max_width) return fail(PQD_BAD_ARG, "bit width out of range");
  if (n_values & 7) return fail(PQD_BAD_ARG, "value count must be a multiple of 8");
  *total = n_values / 8 * (size_t)bit_width;
  if (in_len < *total) return fail(PQD_EOF, "bit-packed run shorter than n_values*bit_width/8 bytes");
  if (n_values && !out) return fail(PQD_BAD_ARG, "null output");
  if (*total && !in) return fail(PQD_BAD_ARG, "null input");
  return PQD_OK;
}

int pqd_unpack_i32_dev(int bit_width, const uint8_t* in, size_t in_len, int32_t* out, size_t n_values,
                       void* stream) {
  size_t total;
  int st = check_unpack_args(bit_width, 32, in, in_len, out, n_values, &total);
  if (st) return st;
  if (n_values == 0) return PQD_OK;
  if (bit_width == 0) {  // the Java unpack writes nothing; a fresh int[] holds zeros
    PQD_HIP(hipMemsetAsync(out, 0, n_values * sizeof(int32_t), as_stream(stream)));
    return PQD_OK;
  }
  PQD_HIP(pqd::launch_unpack_run_i32(in, total, out, n_values, (uint32_t)bit_width, as_stream(stream)));
  return PQD_OK;
}

int pqd_unpack_i64_dev(int bit_width, const uint8_t* in, size_t in_len, int64_t* out, size_t n_values,
                       void* stream) {
  size_t total;
  int st = check_unpack_args(bit_width, 64, in, in_len, out, n_values, &total);
  if (st) return st;
  if (n_values == 0) return PQD_OK;
  if (bit_width == 0) {
    PQD_HIP(hipMemsetAsync(out, 0, n_values * sizeof(int64_t), as_stream(stream)));
    return PQD_OK;
  }
  PQD_HIP(pqd::launch_unpack_run_i64(in, total, out, n_values, (uint32_t)bit_width, as_stream(stream)));
  return PQD_OK;
}

static int unpack_host(int bit_width, int elem, const uint8_t* in, size_t in_len, void* out, size_t n_values) {
  size_t total;
  int st = check_unpack_args(bit_width, elem == 4 ? 32 : 64, in, in_len, out, n_values, &total);
  if (st) return st;
  if (n_values == 0) return PQD_OK;
  if ((st = check_device_available())) return st;
  HostCtx& c = g_ctx;
  hipStream_t s;
  if ((st = c.ensure_stream(&s))) return st;
  if ((st = c.in.ensure(total + 16))) return st;
  if ((st = c.out.ensure(n_values * (size_t)elem))) return st;
  if (total) PQD_HIP(hipMemcpyAsync(c.in.p, in, total, hipMemcpyHostToDevice, s));
  if (elem == 4)
    st = pqd_unpack_i32_dev(bit_width, (const uint8_t*)c.in.p, total, (int32_t*)c.out.p, n_values, s);
  else
    st = pqd_unpack_i64_dev(bit_width, (const uint8_t*)c.in.p, total, (int64_t*)c.out.p, n_values, s);
  if (st) return st;
  PQD_HIP(hipMemcpyAsync(out, c.out.p, n_values * (size_t)elem, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  return PQD_OK;
}

int pqd_unpack_i32(int bit_width, const uint8_t* in, size_t in_len, int32_t* out, size_t n_values) {
  return unpack_host(bit_width, 4, in, in_len, out, n_values);
}
int pqd_unpack_i64(int bit_width, const uint8_t* in, size_t in_len, int64_t* out, size_t n_values) {
  return unpack_host(bit_width, 8, in, in_len, out, n_values);
}

/* ---- page decode ---------------------------------------------------------------------------- */

int pqd_decode_pages_dev(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int encoding,
                         int elem_size, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  hipStream_t s = as_stream(stream);
  switch (encoding) {
    case PQD_ENC_RLE:
      if (elem_size != 4) return fail(PQD_BAD_ARG, "PQD_ENC_RLE produces 4-byte values");
      if (!use_planned_rle_path()) {
        PQD_HIP(pqd::launch_rle_pages(descs, results, n_pages, 4, false, nullptr, nullptr, s));
      } else {
        // 1. header walk of every page -> chunks of run segments in the pool
        // 2. expansion of all chunks (workgroups take chunks from a counter)
        // 3. generic kernel for the pages the plan left over
        PlanLease lease;  // holds the scratch until the last launch below is enqueued
        int st = plan_scratch_acquire(s, n_pages, 0, &lease);
        if (st) return st;
        PlanScratch& ps = *lease.ps;
        const uint32_t pool_chunks = usable_pool_chunks(ps.gen_chunks);
        uint32_t *counters, *other;
        if ((st = counters_begin(ps, s, &counters, &other))) return st;
        PQD_HIP(pqd::launch_rle_plan_chunks(descs, results, ps.plans, ps.gen_pool, pool_chunks, counters, n_pages, false, 4,
                                            pqd::DictPlanes{nullptr, nullptr, nullptr, 0u}, use_small_rle_kernel(), s,
                                            n_pages <= pqd::WP_LONG_MAX_PAGES));
        PQD_HIP(pqd::launch_rle_expand(ps.gen_pool, pool_chunks, counters, n_pages, s));
        PQD_HIP(pqd::launch_rle_pages(descs, results, n_pages, 4, false, ps.plans, counters + 1, s, nullptr, 0, nullptr, other, COUNTER_ZERO_WORDS));
        counters_end(ps);
      }
      return PQD_OK;
    case PQD_ENC_RLE_DICTIONARY:
      if (elem_size != 4 && elem_size != 8) return fail(PQD_BAD_ARG, "dictionary values must be 4 or 8 bytes");
      if (!use_sliced_dictionary_path()) {
        PQD_HIP(pqd::launch_rle_pages(descs, results, n_pages, elem_size, true, nullptr, nullptr, s));
      } else {
        // 1. header walk of every page (one wave per page) -> chunks of run segments in the pool
        // 2. expansion + sliced LDS gather, chunk by chunk
        // 3. generic kernel for the pages the plan left over
        PlanLease lease;  // holds the scratch until the last launch below is enqueued
        // 8-byte dictionaries beyond one LDS slice are swept as dword planes: room for the plane copies
        // of one dictionary per 16 pages (at least 4, at most DP_MAX_SLOTS; PQD_PLANE_SLOTS overrides);
        // dictionaries that find no slot are gathered from memory
        int st = plan_scratch_acquire(s, n_pages, elem_size == 8 ? wanted_plane_slots(n_pages) : 0u, &lease);
        if (st) return st;
        PlanScratch& ps = *lease.ps;
        uint32_t *counters, *other;
        if ((st = counters_begin(ps, s, &counters, &other))) return st;
        const pqd::DictPlanes planes = planes_of(counters, ps.plane_area, elem_size == 8 ? usable_plane_slots(ps.plane_slots, n_pages) : 0u);
        if ((st = phase_mark(0, s))) return st;
        const uint32_t pool_chunks = usable_pool_chunks(ps.gen_chunks);
        PQD_HIP(pqd::launch_rle_plan_chunks(descs, results, ps.plans, ps.gen_pool, pool_chunks, counters, n_pages, true,
                                            elem_size, planes, false, s, n_pages <= pqd::WP_LONG_MAX_PAGES));
        if ((st = phase_mark(1, s))) return st;
        // (the sliced gather takes the page slots and the front half of the pool's tail; the back half holds the
        // chunks that the left-over launch expands)
        hipEvent_t ev_start, ev_stop;
        phase_gather_events(&ev_start, &ev_stop);
        PQD_HIP(pqd::launch_dict_pages_sliced(results, ps.gen_pool, pqd::plan_front_limit(pool_chunks, n_pages, true), counters, n_pages,
                                              elem_size, planes.area, s, ev_start, ev_stop));
        if ((st = phase_mark(2, s))) return st;
        PQD_HIP(pqd::launch_rle_pages(descs, results, n_pages, elem_size, true, ps.plans, counters + 1, s, ps.gen_pool, pool_chunks, counters, other,
                                      COUNTER_ZERO_WORDS));
        counters_end(ps);
        if ((st = phase_mark(3, s))) return st;
      }
      return PQD_OK;
    case PQD_ENC_DELTA_BINARY_PACKED:
      if (elem_size != 4 && elem_size != 8) return fail(PQD_BAD_ARG, "delta values must be 4 or 8 bytes");
      PQD_HIP(pqd::launch_delta_pages(descs, results, n_pages, elem_size, s));
      return PQD_OK;
    case PQD_ENC_BYTE_STREAM_SPLIT:
      PQD_HIP(pqd::launch_bss_pages(descs, results, n_pages, s));
      return PQD_OK;
    case PQD_ENC_BIT_PACKED:
      if (elem_size != 4) return fail(PQD_BAD_ARG, "PQD_ENC_BIT_PACKED produces 4-byte values");
      PQD_HIP(pqd::launch_bitpacked_pages(descs, results, n_pages, s));
      return PQD_OK;
    case PQD_ENC_PLAIN:
      if (elem_size < 1 || elem_size > 16) return fail(PQD_BAD_ARG, "PLAIN elements are 1..16 bytes");
      PQD_HIP(pqd::launch_plain_pages(descs, results, n_pages, s));
      return PQD_OK;
    case PQD_ENC_PLAIN_BOOLEAN:
      if (elem_size != 1 && elem_size != 4) return fail(PQD_BAD_ARG, "PLAIN booleans decode to 1- or 4-byte values");
      PQD_HIP(pqd::launch_plain_boolean_pages(descs, results, n_pages, elem_size, s));
      return PQD_OK;
    default:
      return fail(PQD_BAD_ARG, "unknown encoding");
  }
}

int pqd_def_levels_to_validity_dev(const pqd_validity_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  PQD_HIP(pqd::launch_def_validity(descs, results, n_pages, as_stream(stream)));
  return PQD_OK;
}

int pqd_scatter_nulls_dev(const pqd_scatter_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem_size,
                          void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  if (elem_size < 1 || elem_size > 16) return fail(PQD_BAD_ARG, "elements are 1..16 bytes");
  PQD_HIP(pqd::launch_scatter_nulls(descs, results, n_pages, elem_size, as_stream(stream)));
  return PQD_OK;
}

int pqd_rep_levels_to_offsets_dev(const pqd_offsets_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  PQD_HIP(pqd::launch_rep_offsets(descs, results, n_pages, as_stream(stream)));
  return PQD_OK;
}

int pqd_decode_pages_dev_noalloc(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int encoding,
                                 int elem_size, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  if (encoding != PQD_ENC_RLE && encoding != PQD_ENC_RLE_DICTIONARY)  // the other encodings use no scratch at all
    return pqd_decode_pages_dev(descs, results, n_pages, encoding, elem_size, stream);
  if (n_pages > NOALLOC_MAX_PAGES)
    return fail(PQD_BAD_ARG, "pqd_decode_pages_dev_noalloc: more than 2048 pages in one call (split the batch, or use pqd_decode_pages_dev)");
  if (encoding == PQD_ENC_RLE && elem_size != 4) return fail(PQD_BAD_ARG, "PQD_ENC_RLE produces 4-byte values");
  if (encoding == PQD_ENC_RLE_DICTIONARY && elem_size != 4 && elem_size != 8)
    return fail(PQD_BAD_ARG, "dictionary values must be 4 or 8 bytes");
  hipStream_t s = as_stream(stream);
  // One launch sequence at a time is enqueued (the mutex), and one at a time runs (the event): a call on
  // another stream starts only when the call before it has left the array.  A stream under graph capture
  // takes no part in that (an event recorded outside the capture cannot be waited for inside it): replays of
  // such a graph must not overlap other calls on the same GPU.
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  unsigned char* base;
  int dev = 0;
  int st = noalloc_scratch(&base, &dev);
  if (st) return st;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  PQD_HIP(hipStreamIsCapturing(s, &cap));
  const bool ordered = cap == hipStreamCaptureStatusNone;
  if (ordered && g_noalloc_recorded[dev]) PQD_HIP(hipStreamWaitEvent(s, g_noalloc_done[dev], 0));
  void* pool = base;
  void* plans = base + NOALLOC_POOL_BYTES;
  uint32_t* counters = reinterpret_cast<uint32_t*>(base + NOALLOC_POOL_BYTES + NOALLOC_PLANS_BYTES);
  const uint32_t pool_chunks = usable_pool_chunks(NOALLOC_POOL_CHUNKS);
  const bool dict = encoding == PQD_ENC_RLE_DICTIONARY;
  const pqd::DictPlanes planes = planes_of(counters, base + NOALLOC_POOL_BYTES + NOALLOC_PLANS_BYTES + NOALLOC_COUNTER_BYTES,
                                           dict && elem_size == 8 ? usable_plane_slots(NOALLOC_PLANE_SLOTS, NOALLOC_PLANE_SLOTS * 16) : 0u);
  // the array keeps the last call's contents: the counters and the plane key table must start at
  // zero; plans, chunks and plane copies are written by the kernels before anything reads them
  PQD_HIP(hipMemsetAsync(counters, 0, COUNTER_BYTES + PLANE_KEYS_BYTES, s));
  PQD_HIP(pqd::launch_rle_plan_chunks(descs, results, plans, pool, pool_chunks, counters, n_pages, dict, elem_size, planes,
                                      use_small_rle_kernel(), s));
  if (dict) {
    PQD_HIP(pqd::launch_dict_pages_sliced(results, pool, pqd::plan_front_limit(pool_chunks, n_pages, true), counters, n_pages, elem_size,
                                          planes.area, s));
  } else
    PQD_HIP(pqd::launch_rle_expand(pool, pool_chunks, counters, n_pages, s));
  PQD_HIP(pqd::launch_rle_pages(descs, results, n_pages, elem_size, dict, plans, counters + 1, s, dict ? pool : nullptr, pool_chunks, counters));
  if (ordered) {
    PQD_HIP(hipEventRecord(g_noalloc_done[dev], s));
    g_noalloc_recorded[dev] = true;
  }
  return PQD_OK;
}

int pqd_decode_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  int st = check_device_available();
  if (st) return st;
  hipStream_t s = as_stream(stream);
  HostCtx& c = g_ctx;
  if ((st = c.descs.ensure((size_t)n_pages * sizeof(pqd_page_desc)))) return st;
  if ((st = c.results.ensure((size_t)n_pages * sizeof(pqd_page_result)))) return st;
  // order the pages by launch class, remembering where each came from
  std::vector<uint32_t> order(n_pages);
  for (uint32_t i = 0; i < n_pages; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(),
                   [&](uint32_t a, uint32_t b) { return class_of(descs[a]) < class_of(descs[b]); });
  std::vector<pqd_page_desc> sorted(n_pages);
  for (uint32_t i = 0; i < n_pages; ++i) sorted[i] = descs[order[i]];
  PQD_HIP(hipMemcpyAsync(c.descs.p, sorted.data(), (size_t)n_pages * sizeof(pqd_page_desc), hipMemcpyHostToDevice, s));
  pqd_page_desc* d_descs = (pqd_page_desc*)c.descs.p;
  pqd_page_result* d_results = (pqd_page_result*)c.results.p;
  for (uint32_t a = 0; a < n_pages;) {
    uint32_t b = a + 1;
    while (b < n_pages && class_of(sorted[b]) == class_of(sorted[a])) ++b;
    st = pqd_decode_pages_dev(d_descs + a, d_results + a, b - a, sorted[a].encoding, sorted[a].elem_size, s);
    if (st == PQD_BAD_ARG) {
      // an encoding class the library does not know: report it per page, keep decoding the others
      std::vector<pqd_page_result> bad(b - a, pqd_page_result{PQD_BAD_ARG, 0u, 0u, 0u});
      PQD_HIP(hipMemcpyAsync(d_results + a, bad.data(), bad.size() * sizeof(pqd_page_result),
                             hipMemcpyHostToDevice, s));
      PQD_HIP(hipStreamSynchronize(s));
    } else if (st) {
      return st;
    }
    a = b;
  }
  std::vector<pqd_page_result> tmp(n_pages);
  PQD_HIP(hipMemcpyAsync(tmp.data(), d_results, (size_t)n_pages * sizeof(pqd_page_result), hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  for (uint32_t i = 0; i < n_pages; ++i) results[order[i]] = tmp[i];
  return PQD_OK;
}

int pqd_decode_page_flat(int encoding, int elem_size, int bit_width, uint32_t flags, const uint8_t* data,
                         uint32_t data_len, uint32_t num_values, const void* dict, uint32_t dict_len, void* out,
                         pqd_page_result* result) {
  if (!result) return fail(PQD_BAD_ARG, "null result");
  if (encoding < 0 || encoding > 255 || elem_size < 0 || elem_size > 255 || bit_width < 0 || bit_width > 255 || flags > 255u)
    return fail(PQD_BAD_ARG, "encoding / element size / bit width / flags out of range");
  int st = check_device_available();
  if (st) return st;
  HostCtx& c = g_ctx;
  hipStream_t s;
  if ((st = c.ensure_stream(&s))) return st;
  pqd_page_desc d{};
  d.data_len = data_len;
  d.num_values = num_values;
  d.dict_len = dict_len;
  d.encoding = (uint8_t)encoding;
  d.elem_size = (uint8_t)elem_size;
  d.bit_width = (uint8_t)bit_width;
  d.flags = (uint8_t)flags;
  const size_t out_bytes = (size_t)num_values * (elem_size ? (size_t)elem_size : 1);
  if (data_len && !data) return fail(PQD_BAD_ARG, "null page data");
  if (out_bytes && !out) return fail(PQD_BAD_ARG, "null output");
  if ((st = c.in.ensure((size_t)data_len + 16))) return st;
  if ((st = c.out.ensure(out_bytes + 16))) return st;
  if (data_len) PQD_HIP(hipMemcpyAsync(c.in.p, data, data_len, hipMemcpyHostToDevice, s));
  d.data = (const uint8_t*)c.in.p;
  d.out = c.out.p;
  if (encoding == PQD_ENC_RLE_DICTIONARY) {
    const size_t dict_bytes = (size_t)dict_len * (size_t)elem_size;
    if (dict_bytes && !dict) return fail(PQD_BAD_ARG, "null dictionary");
    if ((st = c.dict.ensure(dict_bytes + 16))) return st;
    if (dict_bytes) PQD_HIP(hipMemcpyAsync(c.dict.p, dict, dict_bytes, hipMemcpyHostToDevice, s));
    d.dict = c.dict.p;
  }
  st = pqd_decode_pages(&d, result, 1, s);
  if (st) return st;
  const size_t valid = (size_t)result->values_written * (size_t)elem_size;
  if (valid) PQD_HIP(hipMemcpyAsync(out, c.out.p, valid, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  return PQD_OK;
}

int pqd_decode_page(const pqd_page_desc* desc, pqd_page_result* result) {
  if (!desc || !result) return fail(PQD_BAD_ARG, "null descriptor / result");
  return pqd_decode_page_flat(desc->encoding, desc->elem_size, desc->bit_width, desc->flags, desc->data, desc->data_len,
                              desc->num_values, desc->dict, desc->dict_len, desc->out, result);
}

int pqd_thread_release(void) {
  g_ctx.release();
  for (PhaseTimer& t : g_phase_series) {
    for (auto e : t.events) (void)hipEventDestroy(e);
    t.events.clear();
    t.used = 0;
    t.device = -1;
  }
  return PQD_OK;
}

/* ---- dictionaries --------------------------------------------------------------------------- */

int pqd_binary_dict_views(const uint8_t* page, size_t page_len, uint32_t n_entries, int fixed_length,
                          uint64_t* views) {
  if (fixed_length < 0) return fail(PQD_BAD_ARG, "Invalid byte array length");
  if ((page_len && !page) || (n_entries && !views)) return fail(PQD_BAD_ARG, "null dictionary page / views");
  int64_t offset = 0;  // a Java int in the reference: a negative length walks backwards
  for (uint32_t i = 0; i < n_entries; ++i) {
    int32_t len = fixed_length;
    if (fixed_length == 0) {
      if (offset < 0 || (uint64_t)offset + 4 > page_len) return fail(PQD_DECODE, "dictionary entry length outside the page");
      uint32_t raw;
      memcpy(&raw, page + offset, 4);
      len = (int32_t)raw;
      offset += 4;
    }
    if (len < 0 || offset < 0 || (uint64_t)offset + (uint64_t)len > page_len || offset > 0x7fffffff)
      views[i] = PQD_VIEW_INVALID;
    else
      views[i] = ((uint64_t)(uint32_t)len << 32) | (uint32_t)offset;
    offset += len;
    if (offset > 0x7fffffffLL || offset < -0x80000000LL) return fail(PQD_DECODE, "dictionary page offsets overflow");
  }
  return PQD_OK;
}

}  // extern "C"

struct pqd_dict {
  void* entries = nullptr;
  uint8_t* bytes = nullptr;
  void* slots = nullptr;  // BINARY kinds: one 16-byte slot per entry for the one-pass decode (k_binary_fused.hip)
  uint32_t n_entries = 0, elem_size = 0, bytes_len = 0, physical_type = 0;
  int device = 0;
};

extern "C" {

int pqd_dict_create(int physical_type, int type_length, const uint8_t* page, size_t page_len, uint32_t n_entries,
                    pqd_dict** dict) {
  if (!dict) return fail(PQD_BAD_ARG, "null dictionary handle");
  *dict = nullptr;
  if (page_len && !page) return fail(PQD_BAD_ARG, "null dictionary page");
  if (page_len > 0x7fffffffu) return fail(PQD_BAD_ARG, "dictionary page larger than a Java array");
  int st = check_device_available();
  if (st) return st;
  pqd_dict d;
  d.physical_type = (uint32_t)physical_type;
  d.n_entries = n_entries;
  PQD_HIP(hipGetDevice(&d.device));
  std::vector<uint64_t> views;
  const void* host_entries = page;
  size_t entry_bytes = 0;
  switch (physical_type) {
    case PQD_TYPE_INT32:
    case PQD_TYPE_FLOAT:
      d.elem_size = 4;
      break;
    case PQD_TYPE_INT64:
    case PQD_TYPE_DOUBLE:
      d.elem_size = 8;
      break;
    case PQD_TYPE_BYTE_ARRAY:
    case PQD_TYPE_FIXED_LEN_BYTE_ARRAY:
    case PQD_TYPE_INT96: {
      int fixed = physical_type == PQD_TYPE_BYTE_ARRAY ? 0 : physical_type == PQD_TYPE_INT96 ? 12 : type_length;
      if (physical_type == PQD_TYPE_FIXED_LEN_BYTE_ARRAY && fixed <= 0) return fail(PQD_BAD_ARG, "Invalid byte array length");
      views.resize(n_entries ? n_entries : 1);
      if ((st = pqd_binary_dict_views(page, page_len, n_entries, fixed, views.data()))) return st;
      d.elem_size = 8;
      host_entries = views.data();
      break;
    }
    default:
      return fail(PQD_BAD_ARG, "no dictionary for this physical type");
  }
  entry_bytes = (size_t)n_entries * d.elem_size;
  if (host_entries == page && entry_bytes > page_len)
    return fail(PQD_DECODE, "dictionary page holds fewer values than its dictionarySize");
  // 16 spare bytes behind both arrays: the kernels read whole 16-byte pieces
  PQD_HIP(hipMalloc(&d.entries, entry_bytes + 16));
  hipError_t e = hipSuccess;
  if (entry_bytes) e = hipMemcpy(d.entries, host_entries, entry_bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess && host_entries != page) {
    d.bytes_len = (uint32_t)page_len;
    e = hipMalloc((void**)&d.bytes, page_len + 16);
    if (e == hipSuccess && page_len) e = hipMemcpy(d.bytes, page, page_len, hipMemcpyHostToDevice);
    // slots: an entry of up to 12 bytes with its length behind it; a longer one as {offset, length, 0, 2^31};
    // an entry that fails when it is read (PQD_VIEW_INVALID) as all ones
    std::vector<uint32_t> slots((size_t)(n_entries ? n_entries : 1) * 4, 0u);
    for (uint32_t i = 0; i < n_entries; ++i) {
      uint32_t* sl = &slots[(size_t)i * 4];
      const uint64_t v = views[i];
      const uint32_t off = (uint32_t)v, len = (uint32_t)(v >> 32);
      if (v == PQD_VIEW_INVALID) {
        sl[0] = sl[1] = sl[2] = sl[3] = 0xFFFFFFFFu;
      } else if (len <= 12u) {
        memcpy(sl, page + off, len);
        sl[3] = len;
      } else {
        sl[0] = off;
        sl[1] = len;
        sl[3] = 0x80000000u;
      }
    }
    if (e == hipSuccess) e = hipMalloc(&d.slots, slots.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(d.slots, slots.data(), slots.size() * 4, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    (void)hipFree(d.entries);
    if (d.bytes) (void)hipFree(d.bytes);
    if (d.slots) (void)hipFree(d.slots);
    return fail(PQD_HIP_ERROR, "dictionary upload", e);
  }
  *dict = new pqd_dict(d);
  return PQD_OK;
}

int pqd_dict_get(const pqd_dict* dict, pqd_dict_info* info) {
  if (!dict || !info) return fail(PQD_BAD_ARG, "null dictionary handle / info");
  info->entries = dict->entries;
  info->bytes = dict->bytes;
  info->n_entries = dict->n_entries;
  info->elem_size = dict->elem_size;
  info->bytes_len = dict->bytes_len;
  info->physical_type = dict->physical_type;
  return PQD_OK;
}

int pqd_dict_destroy(pqd_dict* dict) {
  if (!dict) return PQD_OK;
  hipError_t e1 = hipFree(dict->entries), e2 = dict->bytes ? hipFree(dict->bytes) : hipSuccess;
  if (dict->slots) (void)hipFree(dict->slots);
  delete dict;
  if (e1 != hipSuccess) return fail(PQD_HIP_ERROR, "hipFree(dictionary entries)", e1);
  if (e2 != hipSuccess) return fail(PQD_HIP_ERROR, "hipFree(dictionary bytes)", e2);
  return PQD_OK;
}

int pqd_binary_materialize_dev(const pqd_binary_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                               void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  PQD_HIP(pqd::launch_binary_pages(descs, results, n_pages, as_stream(stream)));
  return PQD_OK;
}

int pqd_decode_binary_pages_dev(const pqd_dict* dict, const pqd_page_desc* pages, const pqd_binary_page_desc* bins,
                                pqd_page_result* page_results, pqd_page_result* bin_results, uint32_t n_pages, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!dict || !pages || !bins || !page_results || !bin_results) return fail(PQD_BAD_ARG, "null dictionary / descriptor / result array");
  if (!dict->slots) return fail(PQD_BAD_ARG, "not a BINARY dictionary");
  hipStream_t s = as_stream(stream);
  int st;
  if ((st = phase_mark(0, s))) return st;
  if ((st = phase_mark(1, s))) return st;
  hipEvent_t ev_start, ev_stop;
  phase_gather_events(&ev_start, &ev_stop);
  PQD_HIP(pqd::launch_binary_fused(pages, bins, dict->slots, dict->n_entries, page_results, bin_results, n_pages, s, ev_start, ev_stop));
  if ((st = phase_mark(2, s))) return st;
  if ((st = phase_mark(3, s))) return st;
  return PQD_OK;
}

int pqd_delta_byte_array_dev(const pqd_delta_bytes_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  PQD_HIP(pqd::launch_delta_length_blocks(descs, n_pages, as_stream(stream)));
  PQD_HIP(pqd::launch_delta_bytes(descs, results, n_pages, as_stream(stream)));
  PQD_HIP(pqd::launch_plain_byte_array(descs, results, n_pages, as_stream(stream)));
  return PQD_OK;
}

int pqd_decode_delta_byte_array_page(int encoding, const uint8_t* page, uint32_t page_len, uint32_t num_values,
                                     const uint8_t* previous, uint32_t previous_len, int32_t* offsets, uint8_t* bytes,
                                     uint32_t bytes_capacity, pqd_page_result* result, uint64_t* bytes_needed) {
  if (!result || !offsets) return fail(PQD_BAD_ARG, "null result / offsets");
  if (encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY && encoding != PQD_ENC_DELTA_BYTE_ARRAY && encoding != PQD_ENC_PLAIN_BYTE_ARRAY)
    return fail(PQD_BAD_ARG, "not a BYTE_ARRAY value encoding (PLAIN_BYTE_ARRAY, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY are)");
  if ((page_len && !page) || (bytes_capacity && !bytes) || (previous_len && !previous))
    return fail(PQD_BAD_ARG, "null page / output / previous value");
  int st = check_device_available();
  if (st) return st;
  HostCtx& c = g_ctx;
  hipStream_t s;
  if ((st = c.ensure_stream(&s))) return st;
  const bool chained = encoding == PQD_ENC_DELTA_BYTE_ARRAY && previous_len != 0;
  // device staging: page | previous value | its two offsets (in `in`), work area (in `out`),
  // offsets + bytes (in `dict`), descriptor, result
  const size_t page_padded = ((size_t)page_len + 15) & ~(size_t)15, prev_padded = ((size_t)previous_len + 15) & ~(size_t)15;
  const size_t offs_padded = (((size_t)num_values + 1) * 4 + 15) & ~(size_t)15;
  if ((st = c.in.ensure(page_padded + prev_padded + 32))) return st;
  if ((st = c.out.ensure(PQD_DELTA_BYTES_WORK(num_values) + 16))) return st;
  if ((st = c.descs.ensure(sizeof(pqd_delta_bytes_desc)))) return st;
  if ((st = c.results.ensure(sizeof(pqd_page_result)))) return st;
  uint8_t* d_in = (uint8_t*)c.in.p;
  if (page_len) PQD_HIP(hipMemcpyAsync(d_in, page, page_len, hipMemcpyHostToDevice, s));
  const int32_t prev_offsets[2] = {0, (int32_t)previous_len};
  if (chained) {
    PQD_HIP(hipMemcpyAsync(d_in + page_padded, previous, previous_len, hipMemcpyHostToDevice, s));
    PQD_HIP(hipMemcpyAsync(d_in + page_padded + prev_padded, prev_offsets, sizeof prev_offsets, hipMemcpyHostToDevice, s));
  }
  pqd_delta_bytes_desc d{};
  d.data = d_in;
  d.work = (uint8_t*)c.out.p;
  d.data_len = page_len;
  d.num_values = num_values;
  d.encoding = (uint32_t)encoding;
  if (chained) {
    d.prev_bytes = d_in + page_padded;
    d.prev_offsets = (const int32_t*)(d_in + page_padded + prev_padded);
    d.prev_count = 1;
  }
  uint64_t need = 0;
  if (encoding == PQD_ENC_PLAIN_BYTE_ARRAY) {
    // a pass that leaves the bytes where they are: the room the values need
    if ((st = c.dict.ensure(offs_padded + 32))) return st;
    d.offsets = (int32_t*)c.dict.p;
    PQD_HIP(hipMemcpyAsync(c.descs.p, &d, sizeof d, hipMemcpyHostToDevice, s));
    PQD_HIP(pqd::launch_plain_byte_array((const pqd_delta_bytes_desc*)c.descs.p, (pqd_page_result*)c.results.p, 1, s));
    pqd_page_result in_place = {};
    PQD_HIP(hipMemcpyAsync(&in_place, c.results.p, sizeof in_place, hipMemcpyDeviceToHost, s));
    PQD_HIP(hipStreamSynchronize(s));
    need = in_place.reserved;
  } else {
    // the length blocks first: their sum is the room the values need
    PQD_HIP(hipMemcpyAsync(c.descs.p, &d, sizeof d, hipMemcpyHostToDevice, s));
    PQD_HIP(pqd::launch_delta_length_blocks((const pqd_delta_bytes_desc*)c.descs.p, 1, s));
    const int blocks = encoding == PQD_ENC_DELTA_BYTE_ARRAY ? 2 : 1;
    std::vector<int32_t> lengths((size_t)num_values * blocks);
    pqd_page_result block_result[2] = {};
    PQD_HIP(hipMemcpyAsync(block_result, c.out.p, sizeof block_result, hipMemcpyDeviceToHost, s));
    if (!lengths.empty())
      PQD_HIP(hipMemcpyAsync(lengths.data(), (uint8_t*)c.out.p + 64, lengths.size() * 4, hipMemcpyDeviceToHost, s));
    PQD_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < blocks; ++b) {
      if (block_result[b].status != PQD_OK) break;  // the page fails before its first value
      const uint32_t have = block_result[b].values_written < num_values ? block_result[b].values_written : num_values;
      for (uint32_t i = 0; i < have; ++i) {
        const int32_t v = lengths[(size_t)b * num_values + i];
        need += v > 0 ? (uint64_t)v : 0u;
      }
    }
  }
  if (bytes_needed) *bytes_needed = need;
  const uint64_t room = need < bytes_capacity ? need : bytes_capacity;
  if ((st = c.dict.ensure(offs_padded + room + 32))) return st;
  d.offsets = (int32_t*)c.dict.p;
  d.bytes = (uint8_t*)c.dict.p + offs_padded;
  d.bytes_capacity = (uint32_t)room;
  PQD_HIP(hipMemcpyAsync(c.descs.p, &d, sizeof d, hipMemcpyHostToDevice, s));
  if (encoding == PQD_ENC_PLAIN_BYTE_ARRAY)
    PQD_HIP(pqd::launch_plain_byte_array((const pqd_delta_bytes_desc*)c.descs.p, (pqd_page_result*)c.results.p, 1, s));
  else
    PQD_HIP(pqd::launch_delta_bytes((const pqd_delta_bytes_desc*)c.descs.p, (pqd_page_result*)c.results.p, 1, s));
  PQD_HIP(hipMemcpyAsync(result, c.results.p, sizeof *result, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  PQD_HIP(hipMemcpyAsync(offsets, d.offsets, ((size_t)result->values_written + 1) * 4, hipMemcpyDeviceToHost, s));
  if (result->reserved) PQD_HIP(hipMemcpyAsync(bytes, d.bytes, result->reserved, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  return PQD_OK;
}

int pqd_codec_supported(int codec) {
  return codec == PQD_CODEC_UNCOMPRESSED || codec == PQD_CODEC_SNAPPY || codec == PQD_CODEC_GZIP || codec == PQD_CODEC_LZ4_RAW ||
         codec == PQD_CODEC_ZSTD;
}

int pqd_decompress_pages_dev(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec,
                             void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!descs || !results) return fail(PQD_BAD_ARG, "null descriptor / result array");
  if (!pqd_codec_supported(codec)) return fail(PQD_BAD_ARG, "codec not supported (UNCOMPRESSED, SNAPPY, GZIP, LZ4_RAW, ZSTD are)");
  PQD_HIP(pqd::launch_decompress_pages(descs, results, n_pages, codec, as_stream(stream)));
  return PQD_OK;
}

int pqd_chunk_decompress_dev(const uint8_t* chunk, const pqd_chunk_page* pages, uint32_t n_pages, int codec,
                             uint8_t* bodies, size_t bodies_capacity, pqd_chunk_page* uncompressed_pages,
                             pqd_decompress_desc* descs, pqd_page_result* results, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!chunk || !pages || !bodies || !uncompressed_pages || !descs || !results)
    return fail(PQD_BAD_ARG, "null chunk / page table / output");
  if (!pqd_codec_supported(codec)) return fail(PQD_BAD_ARG, "codec not supported (UNCOMPRESSED, SNAPPY, GZIP, LZ4_RAW, ZSTD are)");
  static thread_local std::vector<pqd_decompress_desc> host;
  host.assign((size_t)n_pages * 2, pqd_decompress_desc{});
  size_t at = 0;
  for (uint32_t i = 0; i < n_pages; ++i) {
    const pqd_chunk_page& p = pages[i];
    pqd_chunk_page& u = uncompressed_pages[i];
    u = p;
    const size_t room = ((size_t)p.uncompressed_size + 15) & ~(size_t)15;
    if (at + room > bodies_capacity || at > 0xFFFFFFFFull) return fail(PQD_BAD_ARG, "bodies_capacity too small");
    u.body_offset = (uint32_t)at;
    u.compressed_size = p.uncompressed_size;
    u.is_compressed = 0;
    // V2: the level bytes are stored as they are, only the values behind them are compressed (:232-262)
    uint32_t levels = p.type == PQD_PAGE_DATA_V2 ? p.rep_levels_len + p.def_levels_len : 0u;
    if (levels > p.compressed_size || levels > p.uncompressed_size) return fail(PQD_DECODE, "level bytes beyond the page");
    pqd_decompress_desc& a = host[2 * (size_t)i];
    pqd_decompress_desc& b = host[2 * (size_t)i + 1];
    a.src = chunk + p.body_offset;
    a.dst = bodies + at;
    a.src_len = a.dst_len = levels;
    a.codec = PQD_CODEC_UNCOMPRESSED;
    b.src = chunk + p.body_offset + levels;
    b.dst = bodies + at + levels;
    b.src_len = p.compressed_size - levels;
    b.dst_len = p.uncompressed_size - levels;
    b.codec = (p.type == PQD_PAGE_DATA_V2 && !p.is_compressed) ? (uint32_t)PQD_CODEC_UNCOMPRESSED : (uint32_t)codec;
    at += room;
  }
  hipStream_t s = as_stream(stream);
  PQD_HIP(hipMemcpyAsync(descs, host.data(), host.size() * sizeof(pqd_decompress_desc), hipMemcpyHostToDevice, s));
  PQD_HIP(pqd::launch_decompress_pages(descs, results, 2 * n_pages, codec, s));
  return PQD_OK;
}

int pqd_decompress_page(int codec, const uint8_t* src, uint32_t src_len, uint8_t* dst, uint32_t dst_len,
                        pqd_page_result* result) {
  if (!result) return fail(PQD_BAD_ARG, "null result");
  if ((src_len && !src) || (dst_len && !dst)) return fail(PQD_BAD_ARG, "null source / destination");
  if (!pqd_codec_supported(codec)) return fail(PQD_BAD_ARG, "codec not supported (UNCOMPRESSED, SNAPPY, GZIP, LZ4_RAW, ZSTD are)");
  int st = check_device_available();
  if (st) return st;
  HostCtx& c = g_ctx;
  hipStream_t s;
  if ((st = c.ensure_stream(&s))) return st;
  if ((st = c.in.ensure((size_t)src_len + 16))) return st;
  if ((st = c.out.ensure((size_t)dst_len + 16))) return st;
  if ((st = c.descs.ensure(sizeof(pqd_decompress_desc)))) return st;
  if ((st = c.results.ensure(sizeof(pqd_page_result)))) return st;
  if (src_len) PQD_HIP(hipMemcpyAsync(c.in.p, src, src_len, hipMemcpyHostToDevice, s));
  pqd_decompress_desc d{};
  d.src = (const uint8_t*)c.in.p;
  d.dst = (uint8_t*)c.out.p;
  d.src_len = src_len;
  d.dst_len = dst_len;
  d.codec = (uint32_t)codec;
  PQD_HIP(hipMemcpyAsync(c.descs.p, &d, sizeof d, hipMemcpyHostToDevice, s));
  PQD_HIP(pqd::launch_decompress_pages((const pqd_decompress_desc*)c.descs.p, (pqd_page_result*)c.results.p, 1, codec, s));
  PQD_HIP(hipMemcpyAsync(result, c.results.p, sizeof *result, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  if (result->values_written) PQD_HIP(hipMemcpyAsync(dst, c.out.p, result->values_written, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  return PQD_OK;
}

int pqd_decode_binary_page(const pqd_dict* dict, const uint8_t* page, uint32_t page_len, uint32_t num_values,
                           int32_t* offsets, uint8_t* bytes, uint32_t bytes_capacity, pqd_page_result* result) {
  if (!dict || !result) return fail(PQD_BAD_ARG, "null dictionary handle / result");
  if (!dict->bytes) return fail(PQD_BAD_ARG, "not a BINARY dictionary");
  if ((page_len && !page) || !offsets || (bytes_capacity && !bytes)) return fail(PQD_BAD_ARG, "null page / output");
  int st = check_device_available();
  if (st) return st;
  HostCtx& c = g_ctx;
  hipStream_t s;
  if ((st = c.ensure_stream(&s))) return st;
  // device staging: page | views (in `out`) | offsets + bytes (in `dict`) | descriptors + results
  const size_t views_bytes = (size_t)num_values * 8, offs_bytes = ((size_t)num_values + 1) * 4;
  const size_t offs_padded = (offs_bytes + 15) & ~(size_t)15;
  if ((st = c.in.ensure((size_t)page_len + 16))) return st;
  if ((st = c.out.ensure(views_bytes + 16))) return st;
  if ((st = c.dict.ensure(offs_padded + bytes_capacity + 16))) return st;
  if ((st = c.descs.ensure(sizeof(pqd_page_desc) + sizeof(pqd_binary_page_desc)))) return st;
  if ((st = c.results.ensure(2 * sizeof(pqd_page_result)))) return st;
  if (page_len) PQD_HIP(hipMemcpyAsync(c.in.p, page, page_len, hipMemcpyHostToDevice, s));
  pqd_page_desc d{};
  d.data = (const uint8_t*)c.in.p;
  d.out = c.out.p;
  d.dict = dict->entries;
  d.data_len = page_len;
  d.num_values = num_values;
  d.dict_len = dict->n_entries;
  d.encoding = PQD_ENC_RLE_DICTIONARY;
  d.elem_size = 8;
  pqd_page_result ids_result{};
  if ((st = pqd_decode_pages(&d, &ids_result, 1, s))) return st;
  pqd_binary_page_desc b{};
  b.views = (const uint64_t*)c.out.p;
  b.dict_bytes = dict->bytes;
  b.offsets = (int32_t*)c.dict.p;
  b.bytes = (uint8_t*)c.dict.p + offs_padded;
  b.num_values = ids_result.values_written;  // the values the read loop gets before it fails
  b.dict_bytes_len = dict->bytes_len;
  b.bytes_capacity = bytes_capacity;
  pqd_binary_page_desc* d_b = (pqd_binary_page_desc*)((uint8_t*)c.descs.p + sizeof(pqd_page_desc));
  pqd_page_result* d_r = (pqd_page_result*)c.results.p + 1;
  PQD_HIP(hipMemcpyAsync(d_b, &b, sizeof b, hipMemcpyHostToDevice, s));
  if ((st = pqd_binary_materialize_dev(d_b, d_r, 1, s))) return st;
  pqd_page_result bin_result{};
  PQD_HIP(hipMemcpyAsync(&bin_result, d_r, sizeof bin_result, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  PQD_HIP(hipMemcpyAsync(offsets, b.offsets, ((size_t)bin_result.values_written + 1) * 4, hipMemcpyDeviceToHost, s));
  if (bin_result.bytes_consumed)
    PQD_HIP(hipMemcpyAsync(bytes, b.bytes, bin_result.bytes_consumed, hipMemcpyDeviceToHost, s));
  PQD_HIP(hipStreamSynchronize(s));
  result->status = bin_result.status != PQD_OK ? bin_result.status : ids_result.status;
  result->values_written = bin_result.values_written;
  result->bytes_consumed = ids_result.bytes_consumed;
  result->reserved = bin_result.bytes_consumed;  // bytes written to `bytes`
  return PQD_OK;
}

}  // extern "C"
