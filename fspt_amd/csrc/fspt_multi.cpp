// fspt_multi.cpp - one frame over several devices from ONE host thread (include/fspt.h: fspt_multi_*): every
// device traces every n-th 32x32 tile, the read-out gathers the tiles on devices[0].
#include "fspt_internal.hpp"

#include <array>
#include <dlfcn.h>

// ---------------------------------------------------------------------------
// RCCL, loaded on demand (fspt_multi_set_exchange): libfspt has no link-time dependency on it - a host that never asks
// for an RCCL exchange never loads it.  The handful of entry points used, with the types of <rccl/rccl.h> (ROCm 7.2).
// ---------------------------------------------------------------------------
namespace {
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                 // 0 = ncclSuccess
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };  // ncclDataType_t / ncclRedOp_t values
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
};
Rccl g_rccl;

// the process's RCCL: one that is already loaded (a PyTorch host brings its own), else the system's
int rccl_load() {
  if (g_rccl.lib) return FSPT_OK;
  const char *env = getenv("FSPT_RCCL_LIB");
  const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void *h = nullptr;
  for (const char *n : names) if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char *n : names) if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fspt_set_error("RCCL exchange: librccl.so could not be loaded (%s); set FSPT_RCCL_LIB", dlerror()); return FSPT_E_STATE; }
  Rccl r;
  r.lib = h;
  bool ok = true;
#define RCCL_SYM(field, name) do { *(void **)(&r.field) = dlsym(h, name); ok = ok && r.field != nullptr; } while (0)
  RCCL_SYM(GetVersion, "ncclGetVersion"); RCCL_SYM(CommInitAll, "ncclCommInitAll"); RCCL_SYM(CommDestroy, "ncclCommDestroy");
  RCCL_SYM(GetErrorString, "ncclGetErrorString"); RCCL_SYM(GroupStart, "ncclGroupStart"); RCCL_SYM(GroupEnd, "ncclGroupEnd");
  RCCL_SYM(Send, "ncclSend"); RCCL_SYM(Recv, "ncclRecv"); RCCL_SYM(Reduce, "ncclReduce");
#undef RCCL_SYM
  if (!ok) { fspt_set_error("RCCL exchange: the loaded librccl lacks an entry point"); return FSPT_E_STATE; }
  g_rccl = r;
  return FSPT_OK;
}
#define NCCL_TRY(expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != 0) { fspt_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); return FSPT_E_HIP; } \
  } while (0)
// first error inside an ncclGroupStart / ncclGroupEnd section (nothing may return in between)
struct GroupErr {
  int code = 0; // 0 none, 1 HIP, 2 RCCL
  int value = 0;
  const char *what = "";
  bool ok() const { return code == 0; }
  void hip(hipError_t e, const char *w) { if (!code && e != hipSuccess) { code = 1; value = (int)e; what = w; } }
  void nccl(ncclResult_t r, const char *w) { if (!code && r != 0) { code = 2; value = r; what = w; } }
  int report() const {
    if (code == 1) fspt_set_error("%s failed inside an RCCL group: %s", what, hipGetErrorString((hipError_t)value));
    else fspt_set_error("%s failed: %s", what, g_rccl.GetErrorString(value));
    return FSPT_E_HIP;
  }
};
} // namespace

extern "C" {

// ---------------------------------------------------------------------------
// one frame over several devices (include/fspt.h: fspt_multi_*)
// ---------------------------------------------------------------------------
struct fspt_multi {
  uint32_t W = 0, H = 0;
  std::vector<int> devices;
  std::vector<fspt_scene *> scenes;
  std::vector<fspt_target *> targets;
  std::vector<float4 *> packed;   // per target: its own pixels in work-index order (on its device)
  std::vector<float4 *> staging;  // per target: the same, on devices[0]
  std::vector<hipEvent_t> arrived;
  std::vector<int> peer_direct;   // per target: bit 0 = its device can write devices[0]'s memory directly, bit 1 = the reverse
  uint64_t gather_bytes = 0;
  // the read-out exchange (fspt_multi_set_exchange): peer copies, or RCCL over a communicator of the device list
  int exchange = FSPT_EXCHANGE_PEER_COPY;
  std::vector<ncclComm_t> comms;  // RCCL modes: one rank per device (ncclCommInitAll)
  std::vector<float4 *> frame;    // RCCL_REDUCE: per device a full frame that is zero outside its own tiles
  // fspt_multi_last_stage_ms: per device, events around the stages of the most recent read-out - on the device's own
  // stream: [0] before its pack kernel(s), [1] behind them, [2] behind its copy / send / reduce; on devices[0]'s stream:
  // [3] before, [4] behind the scatter of its tiles.  (Created with the handle; recording one costs about a microsecond.)
  std::vector<std::array<hipEvent_t, 5>> stage_ev;
  std::vector<std::array<bool, 5>> stage_set;
};

static int stage_mark(fspt_multi *m, size_t i, int k, hipStream_t s) {
  if (i >= m->stage_ev.size() || !m->stage_ev[i][k]) return FSPT_OK;
  HIP_TRY(hipEventRecord(m->stage_ev[i][k], s));
  m->stage_set[i][k] = true;
  return FSPT_OK;
}
#define STAGE_MARK(i, k, s) do { int rc_ = stage_mark(m, (i), (k), (s)); if (rc_) return rc_; } while (0)

static void multi_pack_params(fspt_target *t, fspt::TilePackP &q) {
  fspt::TraceP tp{};
  fill_trace_params(t, tp);
  q.W = t->W; q.H = t->H; q.vw = t->W; q.vh = t->H; // the read-out moves whole tiles, whatever the viewport
  q.shard = tp.shard; q.n_shards = tp.n_shards; q.tile = tp.tile; q.tiles_x = tp.tiles_x; q.tiles_y = tp.tiles_y;
  q.n_owned_tiles = tp.n_owned_tiles;
}

int fspt_multi_destroy(fspt_multi *m) {
  if (!m) return FSPT_OK;
  for (size_t i = 0; i < m->targets.size(); ++i) {
    if (m->targets[i]) { hipSetDevice(m->devices[i]); hipStreamSynchronize(m->targets[i]->stream); }
  }
  for (size_t i = 0; i < m->devices.size(); ++i) {
    if (i < m->packed.size() && m->packed[i]) { hipSetDevice(m->devices[i]); hipFree(m->packed[i]); }
    if (i < m->staging.size() && m->staging[i]) { hipSetDevice(m->devices[0]); hipFree(m->staging[i]); }
    if (i < m->arrived.size() && m->arrived[i]) { hipSetDevice(m->devices[i]); hipEventDestroy(m->arrived[i]); }
  }
  for (size_t i = 0; i < m->frame.size(); ++i) if (m->frame[i]) { hipSetDevice(m->devices[i]); hipFree(m->frame[i]); }
  for (size_t i = 0; i < m->stage_ev.size(); ++i)
    for (int k = 0; k < 5; ++k)
      if (m->stage_ev[i][k]) { hipSetDevice(m->devices[k < 3 ? i : 0]); hipEventDestroy(m->stage_ev[i][k]); }
  for (ncclComm_t c : m->comms) if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
  for (fspt_target *t : m->targets) fspt_target_destroy(t);
  for (fspt_scene *s : m->scenes) fspt_scene_destroy(s);
  delete m;
  return FSPT_OK;
}

int fspt_multi_create(const fspt_scene_desc *desc, const int *devices, uint32_t n_devices, uint32_t W, uint32_t H, fspt_multi **out) {
  if (!desc || !devices || !out || n_devices == 0 || n_devices > 64) { fspt_set_error("fspt_multi_create: bad argument (1..64 devices)"); return FSPT_E_INVALID; }
  *out = nullptr;
  fspt_multi *m = new fspt_multi();
  m->W = W; m->H = H;
  m->devices.assign(devices, devices + n_devices);
  m->packed.assign(n_devices, nullptr); m->staging.assign(n_devices, nullptr); m->arrived.assign(n_devices, nullptr); m->peer_direct.assign(n_devices, 3);
  m->stage_ev.assign(n_devices, std::array<hipEvent_t, 5>{{nullptr, nullptr, nullptr, nullptr, nullptr}});
  m->stage_set.assign(n_devices, std::array<bool, 5>{{false, false, false, false, false}});
  int rc = FSPT_OK;
  for (uint32_t i = 0; i < n_devices && rc == FSPT_OK; ++i) {
    // one scene copy per DISTINCT device (a device listed twice shares it)
    fspt_scene *s = nullptr;
    for (uint32_t j = 0; j < i; ++j) if (devices[j] == devices[i]) { s = m->targets[j]->scene; break; }
    if (!s) { rc = fspt_scene_create(desc, devices[i], &s); if (rc == FSPT_OK) m->scenes.push_back(s); }
    fspt_target *t = nullptr;
    if (rc == FSPT_OK) rc = fspt_target_create(s, W, H, &t);
    if (rc == FSPT_OK) { m->targets.push_back(t); rc = fspt_target_set_shard(t, i, n_devices, 32); }
    if (rc == FSPT_OK) {
      for (int k = 0; k < 5 && rc == FSPT_OK; ++k) {
        hipError_t e = hipSetDevice(devices[k < 3 ? i : 0]);
        if (e == hipSuccess) e = hipEventCreate(&m->stage_ev[i][k]);
        if (e != hipSuccess) { fspt_set_error("fspt_multi_create: %s", hipGetErrorString(e)); rc = FSPT_E_HIP; }
      }
    }
    if (rc == FSPT_OK && i > 0) {
      fspt::TilePackP q{};
      multi_pack_params(t, q);
      const size_t bytes = (size_t)q.n_owned_tiles * q.tile * q.tile * sizeof(float4);
      hipError_t e = hipSetDevice(devices[i]);
      if (e == hipSuccess) e = hipMalloc((void **)&m->packed[i], bytes ? bytes : 16);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->arrived[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipSetDevice(devices[0]);
      if (e == hipSuccess) e = hipMalloc((void **)&m->staging[i], bytes ? bytes : 16);
      if (e == hipSuccess && devices[i] != devices[0]) {
        // The gather copy is issued on the SENDING device's stream and writes devices[0]'s memory (multi_gather), so
        // the mapping that matters is devices[i] -> devices[0]; the reverse one is enabled too (hipMemcpyPeerAsync may
        // pick either end's copy engine).  Without peer access the copy still works, staged through the host.
        int can_out = 0, can_in = 0;
        (void)hipDeviceCanAccessPeer(&can_out, devices[i], devices[0]);
        (void)hipDeviceCanAccessPeer(&can_in, devices[0], devices[i]);
        if (can_out && hipSetDevice(devices[i]) == hipSuccess && hipDeviceEnablePeerAccess(devices[0], 0) != hipSuccess) (void)hipGetLastError(); // already enabled
        if (can_in && hipSetDevice(devices[0]) == hipSuccess && hipDeviceEnablePeerAccess(devices[i], 0) != hipSuccess) (void)hipGetLastError();
        m->peer_direct[i] = (can_out ? 1 : 0) | (can_in ? 2 : 0);
      } else if (e == hipSuccess) {
        m->peer_direct[i] = 3; // the same device
      }
      if (e != hipSuccess) { fspt_set_error("fspt_multi_create: %s", hipGetErrorString(e)); rc = FSPT_E_HIP; }
    }
  }
  if (rc != FSPT_OK) { fspt_multi_destroy(m); return rc; }
  *out = m;
  return FSPT_OK;
}

int fspt_multi_target(fspt_multi *m, uint32_t i, fspt_target **out) {
  if (!m || !out || i >= m->targets.size()) { fspt_set_error("fspt_multi_target: bad argument"); return FSPT_E_INVALID; }
  *out = m->targets[i];
  return FSPT_OK;
}

#define MULTI_EACH(call)                                                     \
  do {                                                                       \
    if (!m) { fspt_set_error("fspt_multi: NULL handle"); return FSPT_E_INVALID; } \
    for (fspt_target *t : m->targets) { int rc_ = (call); if (rc_) return rc_; }  \
    return FSPT_OK;                                                          \
  } while (0)

int fspt_multi_camera(fspt_multi *m, const float P[3], const float I[3], float fov_scale, const float lens[2], float rand_base) {
  MULTI_EACH(fspt_camera(t, P, I, fov_scale, lens, rand_base));
}
int fspt_multi_trace(fspt_multi *m, uint32_t tick, float rand_base, float env_theta, uint32_t num_bounces) {
  MULTI_EACH(fspt_trace(t, tick, rand_base, env_theta, num_bounces));
}
int fspt_multi_render(fspt_multi *m, const fspt_camera_params *cam, uint32_t first_tick, uint32_t n_ticks, uint64_t seed) {
  MULTI_EACH(fspt_render(t, cam, first_tick, n_ticks, seed));
}
int fspt_multi_clear(fspt_multi *m) { MULTI_EACH(fspt_clear(t)); }
int fspt_multi_sync(fspt_multi *m) { MULTI_EACH(fspt_sync(t)); }

int fspt_multi_set_exchange(fspt_multi *m, int mode) {
  if (!m || mode < FSPT_EXCHANGE_PEER_COPY || mode > FSPT_EXCHANGE_RCCL_REDUCE) { fspt_set_error("fspt_multi_set_exchange: bad argument"); return FSPT_E_INVALID; }
  if (mode != FSPT_EXCHANGE_PEER_COPY && m->comms.empty()) {
    // one RCCL rank per device: the devices must be distinct (a device listed twice - the 1-GPU test form - has no
    // communicator; its tiles are on the device already and the peer-copy path handles it)
    for (size_t i = 0; i < m->devices.size(); ++i)
      for (size_t j = 0; j < i; ++j)
        if (m->devices[i] == m->devices[j]) { fspt_set_error("fspt_multi_set_exchange: RCCL needs distinct devices (device %d is listed twice)", m->devices[i]); return FSPT_E_INVALID; }
    int rc = rccl_load();
    if (rc) return rc;
    for (fspt_target *t : m->targets) { int rc_ = fspt_sync(t); if (rc_) return rc_; }
    std::vector<ncclComm_t> comms(m->devices.size(), nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)m->devices.size(), m->devices.data());
    if (r != 0) {
      for (ncclComm_t c : comms) if (c) (void)g_rccl.CommDestroy(c); // whatever it had created before it failed
      fspt_set_error("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
      return FSPT_E_HIP;
    }
    m->comms = comms;
  }
  m->exchange = mode;
  return FSPT_OK;
}

int fspt_multi_get_exchange(fspt_multi *m, int *mode, int *rccl_version) {
  if (!m) { fspt_set_error("fspt_multi_get_exchange: NULL handle"); return FSPT_E_INVALID; }
  if (mode) *mode = m->exchange;
  if (rccl_version) { *rccl_version = 0; if (g_rccl.GetVersion) (void)g_rccl.GetVersion(rccl_version); }
  return FSPT_OK;
}

// RCCL_GATHER: the packed tiles travel as ncclSend / ncclRecv pairs inside one group (RCCL has no gather primitive: this
// is how it spells one), each on its device's stream; devices[0] scatters them behind its receives.
static int multi_gather_rccl(fspt_multi *m) {
  fspt_target *t0 = m->targets[0];
  std::vector<size_t> count(m->targets.size(), 0);
  for (size_t i = 1; i < m->targets.size(); ++i) {
    fspt_target *t = m->targets[i];
    fspt::TilePackP q{};
    multi_pack_params(t, q);
    count[i] = (size_t)q.n_owned_tiles * q.tile * q.tile * 4u; // floats
    if (!count[i]) continue;
    HIP_TRY(hipSetDevice(m->devices[i]));
    q.accum = t->accum; q.packed = m->packed[i];
    STAGE_MARK(i, 0, t->stream);
    HIP_TRY(fspt::launch_tile_pack(q, false, t->stream));
    STAGE_MARK(i, 1, t->stream);
    m->gather_bytes += count[i] * 4u;
  }
  // Inside the group nothing returns: the first error is kept, ncclGroupEnd is ALWAYS called (an open group would stall
  // the host's own later collectives - the library prefers the librccl the process has already loaded), then reported.
  NCCL_TRY(g_rccl.GroupStart());
  GroupErr ge;
  for (size_t i = 1; i < m->targets.size() && ge.ok(); ++i) {
    if (!count[i]) continue;
    ge.hip(hipSetDevice(m->devices[i]), "hipSetDevice");
    if (ge.ok()) ge.nccl(g_rccl.Send(m->packed[i], count[i], NCCL_FLOAT32, 0, m->comms[i], m->targets[i]->stream), "ncclSend");
    if (ge.ok()) ge.hip(hipSetDevice(m->devices[0]), "hipSetDevice");
    if (ge.ok()) ge.nccl(g_rccl.Recv(m->staging[i], count[i], NCCL_FLOAT32, (int)i, m->comms[0], t0->stream), "ncclRecv");
  }
  ge.nccl(g_rccl.GroupEnd(), "ncclGroupEnd");
  if (!ge.ok()) return ge.report();
  for (size_t i = 1; i < m->targets.size(); ++i) {
    if (!count[i]) continue;
    HIP_TRY(hipSetDevice(m->devices[i]));
    STAGE_MARK(i, 2, m->targets[i]->stream);
  }
  HIP_TRY(hipSetDevice(m->devices[0]));
  for (size_t i = 1; i < m->targets.size(); ++i) {
    if (!count[i]) continue;
    fspt::TilePackP q{};
    multi_pack_params(m->targets[i], q);
    q.accum = t0->accum; q.packed = m->staging[i];
    STAGE_MARK(i, 3, t0->stream);
    HIP_TRY(fspt::launch_tile_pack(q, true, t0->stream));
    STAGE_MARK(i, 4, t0->stream);
  }
  return FSPT_OK;
}

// RCCL_REDUCE (what north_star names: "an RCCL reduce of the radiance buffer over xGMI"): every device builds a full
// frame that is zero outside its own tiles (its accumulator may hold other devices' pixels from an earlier read-out on
// devices[0]), ncclReduce(SUM) to devices[0] - every pixel has exactly one owner, so the sum IS the gather, x + 0 + ... + 0
// exactly - and devices[0] takes the result as its accumulator.  n times the bytes of the gather on the links.
static int multi_reduce_rccl(fspt_multi *m) {
  const size_t px = (size_t)m->W * m->H;
  if (m->frame.empty()) m->frame.assign(m->targets.size(), nullptr);
  for (size_t i = 0; i < m->targets.size(); ++i) {
    fspt_target *t = m->targets[i];
    HIP_TRY(hipSetDevice(m->devices[i]));
    fspt::TilePackP q{};
    multi_pack_params(t, q);
    const size_t bytes = (size_t)q.n_owned_tiles * q.tile * q.tile * sizeof(float4);
    if (!m->frame[i]) HIP_TRY(hipMalloc((void **)&m->frame[i], px * sizeof(float4)));
    if (!m->packed[i]) HIP_TRY(hipMalloc((void **)&m->packed[i], bytes ? bytes : 16)); // (devices[0] has none from fspt_multi_create)
    STAGE_MARK(i, 0, t->stream);
    HIP_TRY(hipMemsetAsync(m->frame[i], 0, px * sizeof(float4), t->stream));
    if (bytes) {
      q.accum = t->accum; q.packed = m->packed[i];
      HIP_TRY(fspt::launch_tile_pack(q, false, t->stream));
      q.accum = m->frame[i];
      HIP_TRY(fspt::launch_tile_pack(q, true, t->stream));
    }
    STAGE_MARK(i, 1, t->stream);
    if (i) m->gather_bytes += px * sizeof(float4);
  }
  NCCL_TRY(g_rccl.GroupStart());
  GroupErr ge; // (as in multi_gather_rccl: no return between GroupStart and GroupEnd)
  for (size_t i = 0; i < m->targets.size() && ge.ok(); ++i) {
    ge.hip(hipSetDevice(m->devices[i]), "hipSetDevice");
    if (ge.ok()) ge.nccl(g_rccl.Reduce(m->frame[i], m->frame[i], px * 4u, NCCL_FLOAT32, NCCL_SUM, 0, m->comms[i], m->targets[i]->stream), "ncclReduce");
  }
  ge.nccl(g_rccl.GroupEnd(), "ncclGroupEnd");
  if (!ge.ok()) return ge.report();
  for (size_t i = 0; i < m->targets.size(); ++i) {
    HIP_TRY(hipSetDevice(m->devices[i]));
    STAGE_MARK(i, 2, m->targets[i]->stream);
  }
  HIP_TRY(hipSetDevice(m->devices[0]));
  STAGE_MARK(0, 3, m->targets[0]->stream);
  HIP_TRY(hipMemcpyAsync(m->targets[0]->accum, m->frame[0], px * sizeof(float4), hipMemcpyDeviceToDevice, m->targets[0]->stream));
  STAGE_MARK(0, 4, m->targets[0]->stream);
  return FSPT_OK;
}

// every device packs its own tiles and sends them to devices[0] on its own stream; devices[0] scatters them
static int multi_gather(fspt_multi *m) {
  fspt_target *t0 = m->targets[0];
  m->gather_bytes = 0;
  for (auto &f : m->stage_set) f = {{false, false, false, false, false}};
  for (fspt_target *t : m->targets) FLUSH_OR_RETURN(t); // recorded two-call ticks of every device run before its tiles are packed
  if (m->exchange == FSPT_EXCHANGE_RCCL_GATHER) return multi_gather_rccl(m);
  if (m->exchange == FSPT_EXCHANGE_RCCL_REDUCE) return multi_reduce_rccl(m);
  for (size_t i = 1; i < m->targets.size(); ++i) {
    fspt_target *t = m->targets[i];
    fspt::TilePackP q{};
    multi_pack_params(t, q);
    const size_t bytes = (size_t)q.n_owned_tiles * q.tile * q.tile * sizeof(float4);
    if (!bytes) continue;
    HIP_TRY(hipSetDevice(m->devices[i]));
    q.accum = t->accum; q.packed = m->packed[i];
    STAGE_MARK(i, 0, t->stream);
    HIP_TRY(fspt::launch_tile_pack(q, false, t->stream));
    STAGE_MARK(i, 1, t->stream);
    HIP_TRY(hipMemcpyPeerAsync(m->staging[i], m->devices[0], m->packed[i], m->devices[i], bytes, t->stream));
    STAGE_MARK(i, 2, t->stream);
    HIP_TRY(hipEventRecord(m->arrived[i], t->stream));
    m->gather_bytes += bytes;
  }
  HIP_TRY(hipSetDevice(m->devices[0]));
  for (size_t i = 1; i < m->targets.size(); ++i) {
    fspt::TilePackP q{};
    multi_pack_params(m->targets[i], q);
    if (!q.n_owned_tiles) continue;
    HIP_TRY(hipStreamWaitEvent(t0->stream, m->arrived[i], 0));
    q.accum = t0->accum; q.packed = m->staging[i];
    STAGE_MARK(i, 3, t0->stream);
    HIP_TRY(fspt::launch_tile_pack(q, true, t0->stream));
    STAGE_MARK(i, 4, t0->stream);
  }
  return FSPT_OK;
}

int fspt_multi_read_radiance(fspt_multi *m, float *out) {
  if (!m || !out) { fspt_set_error("fspt_multi_read_radiance: NULL argument"); return FSPT_E_INVALID; }
  int rc = multi_gather(m);
  if (rc) return rc;
  return fspt_read_radiance(m->targets[0], out);
}

int fspt_multi_draw(fspt_multi *m, float exposure, float saturation, int denoise, float max_sigma, uint8_t *out_rgba8) {
  if (!m || !out_rgba8) { fspt_set_error("fspt_multi_draw: NULL argument"); return FSPT_E_INVALID; }
  int rc = multi_gather(m);
  if (rc) return rc;
  return fspt_draw(m->targets[0], exposure, saturation, denoise, max_sigma, out_rgba8);
}

// ---------------------------------------------------------------------------
// The read-out's pack / unpack kernels for hosts that run one process per GPU and move the bytes themselves
// (torch.distributed over RCCL: fspt_amd/distributed.py TileGather, bench.py --gpus N)
// ---------------------------------------------------------------------------
static void shard_pack_params(fspt_target *t, uint32_t shard, uint32_t n_shards, fspt::TilePackP &q) {
  q.W = t->W; q.H = t->H; q.vw = t->W; q.vh = t->H;
  q.shard = shard; q.n_shards = n_shards; q.tile = t->tile;
  q.tiles_x = (t->W + t->tile - 1) / t->tile; q.tiles_y = (t->H + t->tile - 1) / t->tile;
  const uint32_t n_tiles = q.tiles_x * q.tiles_y;
  q.n_owned_tiles = n_tiles > shard ? (n_tiles - shard + n_shards - 1) / n_shards : 0;
}

int fspt_target_shard_slots(fspt_target *t, uint32_t shard, uint32_t n_shards, uint64_t *slots) {
  if (!t || !slots || n_shards == 0 || shard >= n_shards) { fspt_set_error("fspt_target_shard_slots: bad argument"); return FSPT_E_INVALID; }
  fspt::TilePackP q{};
  shard_pack_params(t, shard, n_shards, q);
  *slots = (uint64_t)q.n_owned_tiles * q.tile * q.tile;
  return FSPT_OK;
}

int fspt_target_pack_tiles(fspt_target *t, void *packed_device, uint32_t channels) {
  if (!t || !packed_device || (channels != 3 && channels != 4)) { fspt_set_error("fspt_target_pack_tiles: bad argument (channels 3 or 4)"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  fspt::TilePackP q{};
  shard_pack_params(t, t->shard, t->n_shards, q);
  q.accum = t->accum; q.packed = (float4 *)packed_device; q.channels = channels;
  HIP_TRY(fspt::launch_tile_pack(q, false, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream)); // the caller's collective runs on a stream of its own
  return FSPT_OK;
}

int fspt_target_unpack_tiles(fspt_target *t, const void *packed_device, uint32_t shard, uint32_t n_shards, uint32_t channels) {
  if (!t || !packed_device || (channels != 3 && channels != 4) || n_shards == 0 || shard >= n_shards) { fspt_set_error("fspt_target_unpack_tiles: bad argument"); return FSPT_E_INVALID; }
  FLUSH_OR_RETURN(t);
  HIP_TRY(hipSetDevice(t->scene->device));
  fspt::TilePackP q{};
  shard_pack_params(t, shard, n_shards, q);
  q.accum = t->accum; q.packed = (float4 *)const_cast<void *>(packed_device); q.channels = channels;
  HIP_TRY(fspt::launch_tile_pack(q, true, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return FSPT_OK;
}

int fspt_multi_size(fspt_multi *m, uint32_t *W, uint32_t *H) {
  if (!m || !W || !H) { fspt_set_error("fspt_multi_size: NULL argument"); return FSPT_E_INVALID; }
  *W = m->W; *H = m->H;
  return FSPT_OK;
}

int fspt_multi_peer_access(fspt_multi *m, uint32_t i, int *mask) {
  if (!m || !mask || i >= m->targets.size()) { fspt_set_error("fspt_multi_peer_access: bad argument"); return FSPT_E_INVALID; }
  *mask = m->peer_direct[i];
  return FSPT_OK;
}

int fspt_multi_last_stage_ms(fspt_multi *m, float *ms, uint32_t n_devices) {
  if (!m || !ms || n_devices != m->targets.size()) { fspt_set_error("fspt_multi_last_stage_ms: bad argument (ms[n_devices][4])"); return FSPT_E_INVALID; }
  for (size_t i = 0; i < m->targets.size(); ++i) {
    float *o = ms + 4 * i;
    o[0] = o[1] = o[2] = o[3] = -1.0f;
    fspt_target *t = m->targets[i];
    HIP_TRY(hipSetDevice(m->devices[i]));
    HIP_TRY(hipStreamSynchronize(t->stream));
    uint32_t launches = 0;
    float render = 0.0f;
    if (fspt_last_kernel_ms(t, &render, &launches) == FSPT_OK) o[0] = render;
    const auto &e = m->stage_ev[i];
    const auto &f = m->stage_set[i];
    if (f[0] && f[1]) HIP_TRY(hipEventElapsedTime(&o[1], e[0], e[1]));
    if (f[1] && f[2]) HIP_TRY(hipEventElapsedTime(&o[2], e[1], e[2]));
    if (f[3] && f[4]) {
      HIP_TRY(hipSetDevice(m->devices[0]));
      HIP_TRY(hipStreamSynchronize(m->targets[0]->stream));
      HIP_TRY(hipEventElapsedTime(&o[3], e[3], e[4]));
    }
  }
  return FSPT_OK;
}

int fspt_multi_last_gather_bytes(fspt_multi *m, uint64_t *bytes) {
  if (!m || !bytes) { fspt_set_error("fspt_multi_last_gather_bytes: NULL argument"); return FSPT_E_INVALID; }
  *bytes = m->gather_bytes;
  return FSPT_OK;
}

} // extern "C"
