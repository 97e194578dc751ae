// dtsim_api.hip -- C-ABI entry points of libdtsim.so (include/dtsim.h).
// Host-side only: handle management, table packing/upload, stream-ordered launches.
// There is deliberately NO CPU fallback: without a HIP device dtsim_create fails with
// DTSIM_E_NOGPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dtsim_dev.h"
#include <dlfcn.h>
#include <mutex>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(DTSIM_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

struct ProfSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_;
};

}  // namespace

struct dtsim {
  dtsim_config cfg{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int N = 0;
  // SoA slab
  void* slab = nullptr;
  size_t slab_bytes = 0;
  SimArrays A{};
  // maps
  bool have_maps = false, have_reset = false;
  MapSet M{};
  uint64_t* d_blobs = nullptr;
  DynInit* d_dyn = nullptr;
  std::vector<int> map_n_dyn, map_n_obj;
  // reset / pool staging
  dtsim_init_state* d_states = nullptr;
  uint8_t* d_mask = nullptr;
  dtsim_init_state* d_pool = nullptr;
  int n_pool = 0;
  // actions staging
  void* d_actions = nullptr;
  size_t actions_cap = 0;
  // query staging
  int32_t* d_qenv = nullptr;
  double* d_qpose = nullptr;
  dtsim_probe* d_qout = nullptr;
  dtsim_agent_info* d_agent = nullptr;
  bool rendered = false;          // a render pass has written the per-env cameras (dtsim_draw_lines needs them)
  RenderParams last_R{};          // the parameters of that pass (dtsim_draw_leds: projected triangles, tables); last_segment: it was the segment view
  bool last_segment = false;
  bool leds_ok = false;           // last_R still names live buffers (cleared by dtsim_set_assets / dtsim_set_maps / dtsim_set_distortion_lut, which re-allocate)
  float* d_leds = nullptr;        // dtsim_draw_leds: device copy of the caller's spheres
  int leds_cap = 0;
  float* d_lines = nullptr;       // dtsim_draw_lines: device copy of the caller's segments
  int lines_cap = 0;
  int render_tables = 0;          // dt_launch_render: which env-invariant tables are valid (camera LUT + maps unchanged)
  RenderOverlap overlap{};        // render parts (DTSIM_RENDER_PARTS > 1): second stream + ordering events
  int q_cap = 0;
  // render
  uint8_t* frames_own = nullptr;
  uint8_t* frames = nullptr;
  size_t frames_bytes = 0;
  float* d_lut = nullptr;
  bool have_lut = false;
  uint32_t* d_texels = nullptr;
  uint32_t* d_texels_seg = nullptr;   // segmented versions, same layout as d_texels (dtsim_set_segment_assets)
  uint8_t* d_mesh_seg = nullptr;      // [n_meshes][4] flat segmentation colour per mesh
  TexDev* d_tex = nullptr;
  int n_tex = 0;
  std::vector<TexDev> h_tex;
  MeshDev* d_meshes = nullptr;
  TriDev* d_tris = nullptr;
  int n_meshes = 0;
  std::vector<MeshDev> h_meshes;
  RenderMapDev* d_rmaps = nullptr;
  uint32_t* d_rtiles = nullptr;
  TileLds* d_tilerecs = nullptr;
  ScreenTri* d_stris = nullptr;
  ObjBox* d_objbox = nullptr;
  void* d_objmask = nullptr;    // block boxes [tiles*4][4] floats, then object masks [N][tiles*4] u64
  std::vector<uint32_t> h_pool;       // host copy of the RGBA8 pool (quad blocks are built from it at dtsim_set_maps)
  void* d_pixtab = nullptr;           // per-pixel tables of the shared camera (k_pix_setup)
  uint8_t* d_qtex = nullptr;          // quad-layout blocks for k_raster_q
  uint32_t* d_qtiles = nullptr;
  int n_qtiles = 0, qlog2 = 0;
  int q3_rows = 0;                    // k_raster_v3: LDS table rows (largest padded grid height), 0 = its layout limits are exceeded
  bool raster_old = false;            // DTSIM_RASTER_OLD=1 at dtsim_create: keep k_raster_q (A/B timing only)
  int step_lanes = 1;                 // lanes of a wavefront per env in k_step (physics.hip Coop); DTSIM_STEP_LANES = 1 / 2 / 4 / 8
  float q_per_m = 0.f;
  uint16_t* d_queue = nullptr;
  int32_t* d_qcount = nullptr;
  uint32_t* d_items = nullptr;
  uint16_t* d_qend = nullptr;
  dtsim_reset_sampler* d_sampler = nullptr;   // device copy when a reset sampler is installed
  int map_w[DTSIM_MAX_MAPS] = {0}, map_h[DTSIM_MAX_MAPS] = {0};
  int32_t* d_obsc_tab = nullptr;  // dtsim_observe_cubic tables (device copy of obsc_tab)
  std::vector<int32_t> obsc_tab;
  int32_t* d_obs_tab = nullptr;   // dtsim_observe resampling tables (cached per output size)
  int obs_h = 0, obs_w = 0, obs_kx = 0, obs_ky = 0, obs_rpb = 0, obs_rows_in = 0;
  size_t obs_off_by = 0;
  ObserveParams obs_fast{};       // the power-of-two fast-path fields of the cached output size (hfast .. vw)
  int max_tris = 0;
  int n_tilerecs = 0, tex_w = 1, tex_h = 1;
  ObjInstDev* d_robjs = nullptr;
  void* d_envcam = nullptr;
  ProfSlot prof[DTSIM_KERNEL__COUNT];
};

namespace {

template <typename T>
T* carve(char*& p, size_t count) {
  T* r = reinterpret_cast<T*>(p);
  size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
  p += bytes;
  return r;
}

size_t layout_arrays(SimArrays& A, int N, char* base) {
  char* p = base;
  const size_t n = (size_t)N;
  A.N = N;
  A.pos_x = carve<double>(p, n); A.pos_z = carve<double>(p, n); A.angle = carve<double>(p, n);
  A.q_x = carve<double>(p, n); A.q_y = carve<double>(p, n); A.q_c = carve<double>(p, n); A.q_s = carve<double>(p, n);
  A.vel_u = carve<double>(p, n); A.vel_w = carve<double>(p, n);
  A.ring = carve<double>(p, n * DTSIM_MAX_DELAY * 2);
  A.war = carve<double>(p, n); A.wal = carve<double>(p, n); A.wheel_dist = carve<double>(p, n);
  A.timestamp = carve<double>(p, n); A.speed = carve<double>(p, n); A.reward = carve<double>(p, n);
  A.lane = carve<double>(p, n * 4); A.prox = carve<double>(p, n); A.wheels = carve<double>(p, n * 2);
  A.ob_cx = carve<double>(p, n * DTSIM_MAX_DYNAMIC); A.ob_cz = carve<double>(p, n * DTSIM_MAX_DYNAMIC);
  A.ob_sx = carve<double>(p, n * DTSIM_MAX_DYNAMIC); A.ob_sz = carve<double>(p, n * DTSIM_MAX_DYNAMIC);
  A.ob_corners = carve<double>(p, n * DTSIM_MAX_DYNAMIC * 8);
  A.ob_vel = carve<double>(p, n * DTSIM_MAX_DYNAMIC); A.ob_wait = carve<double>(p, n * DTSIM_MAX_DYNAMIC);
  A.ob_time = carve<double>(p, n * DTSIM_MAX_DYNAMIC); A.ob_angle = carve<double>(p, n * DTSIM_MAX_DYNAMIC);
  A.ob_wiggle = carve<double>(p, n * DTSIM_MAX_DYNAMIC); A.ob_yrot = carve<double>(p, n * DTSIM_MAX_DYNAMIC);
  A.cam = carve<float>(p, n * 6); A.colors = carve<float>(p, n * 16);
  A.ring_head = carve<int32_t>(p, n); A.step_count = carve<int32_t>(p, n);
  A.tile_i = carve<int32_t>(p, n); A.tile_j = carve<int32_t>(p, n);
  A.map_id = carve<int32_t>(p, n); A.episode = carve<int32_t>(p, n);
  A.done = carve<uint8_t>(p, n); A.done_code = carve<uint8_t>(p, n); A.in_lane = carve<uint8_t>(p, n);
  A.ob_active = carve<uint8_t>(p, n * DTSIM_MAX_DYNAMIC);
  A.ob_visible = carve<uint8_t>(p, n * DTSIM_MAX_OBJECTS);
  A.ob_light = carve<uint8_t>(p, n * DTSIM_MAX_OBJECTS);
  A.tl_time = carve<double>(p, n);
  A.ob_cy = carve<double>(p, n * DTSIM_MAX_DYNAMIC);
  A.ob_ext = carve<double>(p, n * DTSIM_MAX_DYNAMIC * 5);
  return (size_t)(p - base);
}

StepParams step_params(const dtsim* h, int n_steps) {
  StepParams P{};
  P.n_steps = n_steps;
  P.frame_skip = h->cfg.frame_skip;
  P.max_steps = h->cfg.max_steps;
  P.delay_steps = h->cfg.delay_steps;
  P.action_mode = h->cfg.action_mode;
  P.actions_f64 = (h->cfg.flags & DTSIM_F_ACTIONS_F64) ? 1 : 0;
  P.auto_reset = ((h->cfg.flags & DTSIM_F_AUTO_RESET) && (h->n_pool > 0 || h->d_sampler)) ? 1 : 0;
  P.n_pool = h->n_pool;
  P.sampler = h->d_sampler;
  P.delta_time = h->cfg.delta_time;
  P.robot_speed = h->cfg.robot_speed;
  P.gain = h->cfg.gain; P.trim = h->cfg.trim; P.radius = h->cfg.radius; P.k = h->cfg.k; P.limit = h->cfg.limit;
  P.lanes = h->step_lanes;
  P.light_capture = (h->cfg.flags & DTSIM_F_LIGHT_CAPTURE) ? 1 : 0;
  P.domain_rand = (h->cfg.flags & DTSIM_F_DOMAIN_RAND) ? 1 : 0;
  return P;
}

struct ProfScope {
  dtsim* h; int k; hipEvent_t a = nullptr, b = nullptr; bool on;
  ProfScope(dtsim* h_, int k_) : h(h_), k(k_), on((h_->cfg.flags & DTSIM_F_PROFILE) != 0) {
    if (!on) return;
    ProfSlot& s = h->prof[k];
    if (!s.free_.empty()) { a = s.free_.back().first; b = s.free_.back().second; s.free_.pop_back(); }
    else { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    (void)hipEventRecord(a, h->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, h->stream);
    h->prof[k].pending.emplace_back(a, b);
  }
};

}  // namespace

extern "C" {

int dtsim_abi_version(void) { return DTSIM_ABI_VERSION; }
const char* dtsim_last_error(void) { return g_err.c_str(); }

int dtsim_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(DTSIM_E_NOGPU, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int dtsim_create(const dtsim_config* cfg, dtsim_t** out) {
  if (!cfg || !out) return fail(DTSIM_E_INVALID, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(dtsim_config))
    return fail(DTSIM_E_INVALID, "dtsim_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(dtsim_config));
  if (cfg->num_envs <= 0) return fail(DTSIM_E_INVALID, "num_envs must be > 0");
  if (cfg->delay_steps < 0 || cfg->delay_steps > DTSIM_MAX_DELAY)
    return fail(DTSIM_E_LIMIT, "delay_steps %d outside [0,%d]", cfg->delay_steps, DTSIM_MAX_DELAY);
  if (cfg->frame_skip < 1 || cfg->delta_time <= 0) return fail(DTSIM_E_INVALID, "bad frame_skip/delta_time");
  if (cfg->cam_width <= 0 || cfg->cam_height <= 0) return fail(DTSIM_E_INVALID, "bad camera size");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(DTSIM_E_NOGPU, "no HIP device visible: libdtsim has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(DTSIM_E_INVALID, "device %d out of range", cfg->device);
  HIPCHK(hipSetDevice(cfg->device));
  dtsim* h = new dtsim();
  { const char* ro = getenv("DTSIM_RASTER_OLD"); h->raster_old = ro && ro[0] == '1'; }   // A/B timing aid: k_raster_q instead of k_raster_v3
  {  // lanes of a wavefront per env in k_step: as many (up to 4) as keep the launch within ~32 K threads -- a small batch is a
     // latency problem (one f64 chain per env, 64 wavefronts on 1024 SIMDs at N = 4096), a large one a throughput problem,
     // where the redundant lanes would cost (profiles/r03_c2_lanes_ab.txt).  DTSIM_STEP_LANES = 1 / 2 / 4 / 8 overrides.
    const char* sl = getenv("DTSIM_STEP_LANES");
    int v = sl ? atoi(sl) : 0;
    if (!(v == 1 || v == 2 || v == 4 || v == 8)) v = cfg->num_envs * 4 <= 32768 ? 4 : cfg->num_envs * 2 <= 32768 ? 2 : 1;
    h->step_lanes = v;
  }
  h->cfg = *cfg;
  h->N = cfg->num_envs;
  if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(DTSIM_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    h->own_stream = true;
  }
  SimArrays tmp{};
  h->slab_bytes = layout_arrays(tmp, h->N, reinterpret_cast<char*>(4096));
  hipError_t e = hipMalloc(&h->slab, h->slab_bytes);
  if (e != hipSuccess) { dtsim_destroy(h); return fail(DTSIM_E_HIP, "hipMalloc(state %zu B): %s", h->slab_bytes, hipGetErrorString(e)); }
  (void)hipMemsetAsync(h->slab, 0, h->slab_bytes, h->stream);
  layout_arrays(h->A, h->N, (char*)h->slab);
  // map_id = -1 everywhere so the first reset creates the world objects
  (void)hipMemsetAsync(h->A.map_id, 0xFF, sizeof(int32_t) * (size_t)h->N, h->stream);
  e = hipMalloc(&h->d_states, sizeof(dtsim_init_state) * (size_t)h->N);
  if (e == hipSuccess) e = hipMalloc(&h->d_mask, (size_t)h->N);
  if (e != hipSuccess) { dtsim_destroy(h); return fail(DTSIM_E_HIP, "hipMalloc(reset staging): %s", hipGetErrorString(e)); }
  if (cfg->flags & DTSIM_F_RENDER) {
    h->frames_bytes = (size_t)h->N * cfg->cam_height * cfg->cam_width * 3;
    e = hipMalloc(&h->frames_own, h->frames_bytes);
    if (e != hipSuccess) { dtsim_destroy(h); return fail(DTSIM_E_HIP, "hipMalloc(frames %zu B): %s", h->frames_bytes, hipGetErrorString(e)); }
    h->frames = h->frames_own;
    e = hipMalloc(&h->d_lut, sizeof(float) * 4 * (size_t)cfg->cam_height * cfg->cam_width);
    if (e == hipSuccess) e = hipMalloc(&h->d_envcam, (size_t)h->N * (128 + 64 + 64 + 4) + 64 + ((size_t)h->N + 1) * 64 + (size_t)h->N * 320);   // EnvCam[N], EnvFast[N], EnvQ[N], render order [N], (aligned) EnvV[N + 1], EnvD[N]
    if (e == hipSuccess) e = hipMalloc(&h->d_pixtab, (size_t)cfg->cam_height * cfg->cam_width * 64 + 2048);   // PixTab + SampTab + 1 KB store dump + debug counters
    {  // MSAA edge queue: one worst-case region per raster wavefront (render.hip QREGION)
      const size_t n_wg = dt_raster_tiles(cfg->cam_width, cfg->cam_height) * (((size_t)h->N + DT_ENVS_PER_BLOCK - 1) / DT_ENVS_PER_BLOCK);
      if (e == hipSuccess) e = hipMalloc(&h->d_queue, n_wg * 4 * (64 * DT_PPT) * DT_ENVS_PER_BLOCK * sizeof(uint16_t));
      if (e == hipSuccess) e = hipMalloc(&h->d_qcount, (n_wg * 4 + 8 + DT_WORK_INTS * DT_MAX_RENDER_PARTS) * sizeof(int32_t));   // counts, debug counters, work-list header (of each render part)
      if (e == hipSuccess) e = hipMalloc(&h->d_items, n_wg * (DT_ITEMS_PER_WG + DT_ENVS_PER_BLOCK) * sizeof(uint32_t));   // k_resolve's list + k_resolve_obj's (at most one per env of a workgroup)
      if (e == hipSuccess) e = hipMalloc(&h->d_qend, n_wg * 4 * DT_ENVS_PER_BLOCK * sizeof(uint16_t));
    }
    if (e != hipSuccess) { dtsim_destroy(h); return fail(DTSIM_E_HIP, "hipMalloc(lut): %s", hipGetErrorString(e)); }
    {  // render parts: off (1) unless asked for
      const char* rp = getenv("DTSIM_RENDER_PARTS");
      const int parts = rp ? std::min(std::max(atoi(rp), 1), DT_MAX_RENDER_PARTS) : 1;
      if (parts > 1) {
        e = hipStreamCreateWithFlags(&h->overlap.s2, hipStreamNonBlocking);
        for (int i = 0; i <= DT_MAX_RENDER_PARTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->overlap.ev[i], hipEventDisableTiming);
        if (e != hipSuccess) { dtsim_destroy(h); return fail(DTSIM_E_HIP, "render parts (stream / events): %s", hipGetErrorString(e)); }
        h->overlap.parts = parts;
      }
    }
    (void)hipMemset((char*)h->d_pixtab + (size_t)cfg->cam_height * cfg->cam_width * 64, 0, 2048);
    if (!(cfg->flags & DTSIM_F_DISTORTION)) {
      // identity LUT: output pixel == rectilinear pixel
      int rc = dtsim_set_distortion_lut(h, nullptr, nullptr);
      if (rc != DTSIM_OK) { dtsim_destroy(h); return rc; }
    }
  }
  *out = h;
  return DTSIM_OK;
}

void dtsim_destroy(dtsim_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& s : h->prof) {
    for (auto& p : s.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto& p : s.free_) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  }
  void* ptrs[] = {h->slab, h->d_blobs, h->d_dyn, h->d_states, h->d_mask, h->d_pool, h->d_actions, h->d_qenv,
                  h->d_qpose, h->d_qout, h->d_agent, h->frames_own, h->d_lut, h->d_texels, h->d_tex, h->d_meshes, h->d_tris,
                  h->d_rmaps, h->d_rtiles, h->d_robjs, h->d_envcam, h->d_tilerecs, h->d_stris, h->d_objbox, h->d_objmask, h->d_queue, h->d_qcount, h->d_items, h->d_qend, h->d_obs_tab, h->d_obsc_tab, h->d_sampler, h->d_texels_seg, h->d_mesh_seg, h->d_qtex, h->d_qtiles, h->d_pixtab, h->d_lines, h->d_leds};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->overlap.s2) { (void)hipStreamSynchronize(h->overlap.s2); (void)hipStreamDestroy(h->overlap.s2); }
  for (hipEvent_t ev : h->overlap.ev) if (ev) (void)hipEventDestroy(ev);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// textures -> one RGBA8 pool: padded (h+1) x (w+1) storage so that GL_REPEAT bilinear fetches never wrap
static int build_texel_pool(const dtsim_texture* textures, int n_textures, std::vector<uint32_t>& pool, std::vector<TexDev>* descs) {
  for (int t = 0; t < n_textures; ++t) {
    const dtsim_texture& tx = textures[t];
    if (tx.width <= 0 || tx.height <= 0 || (tx.width & (tx.width - 1)) || (tx.height & (tx.height - 1)) || !tx.rgba)
      return fail(DTSIM_E_INVALID, "texture %d: size must be a power of two", t);
    TexDev d{tx.width, tx.height, (int32_t)pool.size(), 0};
    const int pw = tx.width + 1;
    pool.resize(pool.size() + (size_t)pw * (tx.height + 1));
    uint32_t* dst = pool.data() + d.off;
    for (int y = 0; y <= tx.height; ++y)
      for (int x = 0; x <= tx.width; ++x) {
        const uint8_t* s = tx.rgba + ((size_t)(y % tx.height) * tx.width + (x % tx.width)) * 4;
        dst[(size_t)y * pw + x] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
      }
    if (descs) descs->push_back(d);
  }
  return DTSIM_OK;
}

int dtsim_set_assets(dtsim_t* h, const dtsim_texture* textures, int n_textures, const dtsim_mesh* meshes,
                     int n_meshes) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  h->leds_ok = false;
  h->render_tables = 0;           // the cached per-pixel / per-block render tables depend on this
  if (n_textures < 0 || n_textures > DTSIM_MAX_TEXTURES) return fail(DTSIM_E_LIMIT, "n_textures %d > %d", n_textures, DTSIM_MAX_TEXTURES);
  if (n_meshes < 0 || n_meshes > DTSIM_MAX_MESHES) return fail(DTSIM_E_LIMIT, "n_meshes %d > %d", n_meshes, DTSIM_MAX_MESHES);
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<uint32_t> pool;
  h->h_tex.clear();
  if (int rc = build_texel_pool(textures, n_textures, pool, &h->h_tex)) return rc;
  h->h_pool = pool;
  // the quad-layout blocks were built from the OLD texel pool (dtsim_set_maps): drop them, so that the generic raster
  // (which reads d_texels) is used until the next dtsim_set_maps rebuilds them -- never a frame mixing both pools
  if (h->d_qtex) { (void)hipFree(h->d_qtex); h->d_qtex = nullptr; }
  if (h->d_qtiles) { (void)hipFree(h->d_qtiles); h->d_qtiles = nullptr; }
  h->n_qtiles = 0; h->qlog2 = 0; h->q3_rows = 0;
  if (h->d_texels_seg) { (void)hipFree(h->d_texels_seg); h->d_texels_seg = nullptr; }   // mirrors the old list
  if (h->d_texels) { (void)hipFree(h->d_texels); h->d_texels = nullptr; }
  if (h->d_tex) { (void)hipFree(h->d_tex); h->d_tex = nullptr; }
  h->n_tex = n_textures;
  if (n_textures > 0) {
    HIPCHK(hipMalloc(&h->d_texels, pool.size() * 4));
    HIPCHK(hipMemcpy(h->d_texels, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_tex, sizeof(TexDev) * n_textures));
    HIPCHK(hipMemcpy(h->d_tex, h->h_tex.data(), sizeof(TexDev) * n_textures, hipMemcpyHostToDevice));
  }
  std::vector<TriDev> tris;
  h->h_meshes.clear();
  for (int m = 0; m < n_meshes; ++m) {
    const dtsim_mesh& ms = meshes[m];
    if (ms.n_tris < 0 || (ms.n_tris > 0 && (!ms.verts || !ms.normals || !ms.colors)))
      return fail(DTSIM_E_INVALID, "mesh %d: null arrays", m);
    MeshDev d{};
    d.n_tris = ms.n_tris; d.off = (int32_t)tris.size();
    for (int k = 0; k < 3; ++k) { d.mn[k] = 1e30f; d.mx[k] = -1e30f; }
    for (int t = 0; t < ms.n_tris * 3; ++t)
      for (int k = 0; k < 3; ++k) { d.mn[k] = std::min(d.mn[k], ms.verts[t * 3 + k]); d.mx[k] = std::max(d.mx[k], ms.verts[t * 3 + k]); }
    for (int t = 0; t < ms.n_tris; ++t) {
      TriDev td;
      memcpy(td.v, ms.verts + (size_t)t * 9, 36);
      memcpy(td.n, ms.normals + (size_t)t * 9, 36);
      memcpy(td.c, ms.colors + (size_t)t * 9, 36);
      if (ms.uvs) memcpy(td.uv, ms.uvs + (size_t)t * 6, 24); else memset(td.uv, 0, 24);
      td.tex = (ms.uvs && ms.tri_tex) ? ms.tri_tex[t] : -1; td.pad = 0;
      if (td.tex >= n_textures) return fail(DTSIM_E_INVALID, "mesh %d triangle %d: texture %d not loaded", m, t, td.tex);
      if (td.tex < 0) td.tex = -1;
      tris.push_back(td);
    }
    h->h_meshes.push_back(d);
  }
  if (h->d_meshes) { (void)hipFree(h->d_meshes); h->d_meshes = nullptr; }
  if (h->d_tris) { (void)hipFree(h->d_tris); h->d_tris = nullptr; }
  h->n_meshes = n_meshes;
  if (n_meshes > 0) {
    HIPCHK(hipMalloc(&h->d_meshes, sizeof(MeshDev) * n_meshes));
    HIPCHK(hipMemcpy(h->d_meshes, h->h_meshes.data(), sizeof(MeshDev) * n_meshes, hipMemcpyHostToDevice));
    if (!tris.empty()) {
      HIPCHK(hipMalloc(&h->d_tris, sizeof(TriDev) * tris.size()));
      HIPCHK(hipMemcpy(h->d_tris, tris.data(), sizeof(TriDev) * tris.size(), hipMemcpyHostToDevice));
    }
  }
  return DTSIM_OK;
}

int dtsim_set_segment_assets(dtsim_t* h, const dtsim_texture* textures, int n_textures, const uint8_t* mesh_rgb, int n_meshes) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  if (n_textures != h->n_tex || n_meshes != h->n_meshes)
    return fail(DTSIM_E_INVALID, "segment assets must mirror dtsim_set_assets (%d textures, %d meshes), got %d / %d", h->n_tex,
                h->n_meshes, n_textures, n_meshes);
  if ((n_textures > 0 && !textures) || (n_meshes > 0 && !mesh_rgb)) return fail(DTSIM_E_INVALID, "null argument");
  for (int t = 0; t < n_textures; ++t)
    if (textures[t].width != h->h_tex[t].w || textures[t].height != h->h_tex[t].h)
      return fail(DTSIM_E_INVALID, "segmented texture %d is %dx%d, the texture it replaces is %dx%d", t, textures[t].width,
                  textures[t].height, h->h_tex[t].w, h->h_tex[t].h);
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<uint32_t> pool;
  if (int rc = build_texel_pool(textures, n_textures, pool, nullptr)) return rc;
  if (h->d_texels_seg) { (void)hipFree(h->d_texels_seg); h->d_texels_seg = nullptr; }
  if (h->d_mesh_seg) { (void)hipFree(h->d_mesh_seg); h->d_mesh_seg = nullptr; }
  HIPCHK(hipMalloc(&h->d_texels_seg, std::max<size_t>(pool.size(), 1) * 4));
  if (!pool.empty()) HIPCHK(hipMemcpy(h->d_texels_seg, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint8_t> rgbx((size_t)std::max(n_meshes, 1) * 4, 0);
  for (int m = 0; m < n_meshes; ++m) { rgbx[m * 4] = mesh_rgb[m * 3]; rgbx[m * 4 + 1] = mesh_rgb[m * 3 + 1]; rgbx[m * 4 + 2] = mesh_rgb[m * 3 + 2]; }
  HIPCHK(hipMalloc(&h->d_mesh_seg, rgbx.size()));
  HIPCHK(hipMemcpy(h->d_mesh_seg, rgbx.data(), rgbx.size(), hipMemcpyHostToDevice));
  return DTSIM_OK;
}

// One quad block (S x S records of 16 B) of a tile texture pre-rotated by the tile angle (render.hip k_raster_q).
// Cell (x0, z0) covers the padded-quad coordinates [x0, x0+1) x [z0, z0+1) of the tile, i.e. the GL_LINEAR taps
// P[z0-1][x0-1], P[z0-1][x0], P[z0][x0-1], P[z0][x0] (GL_REPEAT wrap) of the tile-frame image P with
// P[zz][xx] = T[y][x], (x, y) the texel the tile-local point ((xx+.5)/S, (zz+.5)/S) maps to under glRotatef(angle*90+180)
// and uv = (pu, 1-pv) (simulator.py:394-401,1872-1873): u = {1-fx, fz, fx, 1-fz}[angle], v = {fz, fx, 1-fz, 1-fx}[angle].
// `pool` holds T padded to (S+1) x (S+1).  Meta dword: cells to the nearest tile boundary (DT_QMETA, dtsim_dev.h).
static void build_quad_block(std::vector<uint32_t>& out, const uint32_t* pool, int S, int ang) {
  const size_t base = out.size();
  out.resize(base + (size_t)S * S * 4);
  auto texel = [&](int xx, int zz) -> uint32_t {
    xx &= S - 1; zz &= S - 1;
    int x, y;
    switch (ang & 3) {
      case 0: x = S - 1 - xx; y = zz; break;
      case 1: x = zz; y = xx; break;
      case 2: x = xx; y = S - 1 - zz; break;
      default: x = S - 1 - zz; y = S - 1 - xx; break;
    }
    return pool[(size_t)y * (S + 1) + x];
  };
  for (int z0 = 0; z0 < S; ++z0)
    for (int x0 = 0; x0 < S; ++x0) {
      const uint32_t t00 = texel(x0 - 1, z0 - 1), t10 = texel(x0, z0 - 1), t01 = texel(x0 - 1, z0), t11 = texel(x0, z0);
      // S = 256: 4 x 2 cells per 128-byte line -- record number (x0 >> 2) << 10 | z0 << 2 | (x0 & 3) (render.hip, q8_rec256: a 32 x 2 pixel slot of the
      // raster touches ~ 15 % fewer lines than with the rows of the texture laid end to end); other sizes: row-major
      const size_t rec = (S == 256) ? ((size_t)(x0 >> 2) << 10) | ((size_t)z0 << 2) | (size_t)(x0 & 3) : (size_t)z0 * S + x0;
      uint32_t* q = &out[base + rec * 4];
      for (int c = 0; c < 3; ++c)
        q[c] = ((t00 >> (8 * c)) & 255u) | (((t10 >> (8 * c)) & 255u) << 8) | (((t01 >> (8 * c)) & 255u) << 16) | (((t11 >> (8 * c)) & 255u) << 24);
      q[3] = (uint32_t)std::min(std::min(std::min(x0, S - x0), std::min(z0, S - z0)), 0xFFFF);
    }
}

int dtsim_set_maps(dtsim_t* h, const dtsim_map* maps, int n_maps) {
  if (!h || !maps) return fail(DTSIM_E_INVALID, "null argument");
  h->leds_ok = false;
  h->render_tables = 0;           // the cached per-pixel / per-block render tables depend on this
  if (n_maps <= 0 || n_maps > DTSIM_MAX_MAPS) return fail(DTSIM_E_LIMIT, "n_maps %d outside [1,%d]", n_maps, DTSIM_MAX_MAPS);
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<uint64_t> blobs;
  std::vector<DynInit> dyn((size_t)n_maps * DTSIM_MAX_DYNAMIC);
  memset(dyn.data(), 0, dyn.size() * sizeof(DynInit));
  std::vector<RenderMapDev> rmaps(n_maps);
  std::vector<uint32_t> rtiles;
  std::vector<TileLds> trecs;
  int tex_w = 0, tex_h = 0;
  std::vector<ObjInstDev> robjs;
  MapSet M{};
  M.n_maps = n_maps;
  h->map_n_dyn.assign(n_maps, 0);
  h->map_n_obj.assign(n_maps, 0);
  for (int mi = 0; mi < n_maps; ++mi) {
    const dtsim_map& mp = maps[mi];
    const int nt = mp.grid_w * mp.grid_h;
    if (mp.grid_w <= 0 || mp.grid_h <= 0 || nt > DTSIM_MAX_TILES) return fail(DTSIM_E_LIMIT, "map %d: %d tiles > %d", mi, nt, DTSIM_MAX_TILES);
    if (mp.n_curves < 0 || mp.n_curves > DTSIM_MAX_CURVES) return fail(DTSIM_E_LIMIT, "map %d: n_curves %d", mi, mp.n_curves);
    if (mp.n_objects < 0 || mp.n_objects > DTSIM_MAX_OBJECTS) return fail(DTSIM_E_LIMIT, "map %d: n_objects %d > %d", mi, mp.n_objects, DTSIM_MAX_OBJECTS);
    if (!mp.tile_kind || !mp.tile_angle || !mp.tile_tex || !mp.tile_curve_off || !mp.tile_curve_cnt || !(mp.tile_size > 0))
      return fail(DTSIM_E_INVALID, "map %d: null tile arrays / tile_size", mi);
    if (mp.n_curves > 0 && (!mp.curves || !mp.curve_heads)) return fail(DTSIM_E_INVALID, "map %d: null curves", mi);
    if (mp.n_objects > 0 && !mp.objects) return fail(DTSIM_E_INVALID, "map %d: null objects", mi);
    int n_static = 0, n_dyn = 0;
    for (int o = 0; o < mp.n_objects; ++o) {
      if (mp.objects[o].dynamic) ++n_dyn;
      else if (mp.objects[o].collidable) ++n_static;
    }
    if (n_static > DTSIM_MAX_STATIC) return fail(DTSIM_E_LIMIT, "map %d: %d static collidables > %d", mi, n_static, DTSIM_MAX_STATIC);
    if (n_dyn > DTSIM_MAX_DYNAMIC) return fail(DTSIM_E_LIMIT, "map %d: %d dynamic objects > %d", mi, n_dyn, DTSIM_MAX_DYNAMIC);
    MapHdr hd{};
    h->map_w[mi] = mp.grid_w; h->map_h[mi] = mp.grid_h;
    hd.grid_w = mp.grid_w; hd.grid_h = mp.grid_h; hd.n_curves = mp.n_curves; hd.n_static = n_static;
    hd.n_lights = 0;
    for (int o = 0; o < mp.n_objects; ++o) hd.n_lights += mp.objects[o].light_freq > 0 ? 1 : 0;
    hd.n_dyn = n_dyn; hd.n_obj = mp.n_objects; hd.tile_size = mp.tile_size;
    int w = MAPHDR_WORDS;
    hd.off_tiles = w; w += nt;
    hd.off_curves = w; w += 8 * mp.n_curves;
    hd.off_heads = w; w += 2 * mp.n_curves;
    hd.off_static = w; w += STATIC_WORDS * n_static;
    hd.off_objs = w; w += OBJ_WORDS * mp.n_objects;
    hd.total_words = w;
    const size_t base = blobs.size();
    M.blob_off[mi] = (int32_t)base;
    blobs.resize(base + w);
    uint64_t* b = blobs.data() + base;
    memcpy(b, &hd, sizeof hd);
    for (int t = 0; t < nt; ++t) {
      TileRec tr{};
      tr.kind = mp.tile_kind[t]; tr.angle = mp.tile_angle[t];
      tr.drivable = (tr.kind >= DTSIM_TILE_STRAIGHT && tr.kind <= DTSIM_TILE_4WAY) ? 1 : 0;
      tr.curve_cnt = mp.tile_curve_cnt[t]; tr.curve_off = mp.tile_curve_off[t]; tr.tex = mp.tile_tex[t];
      if (tr.drivable && (tr.curve_off < 0 || tr.curve_off + tr.curve_cnt > mp.n_curves || tr.curve_cnt == 0))
        return fail(DTSIM_E_INVALID, "map %d tile %d: drivable tile without curves", mi, t);
      if (tr.tex >= h->n_tex) return fail(DTSIM_E_INVALID, "map %d tile %d: texture %d not loaded", mi, t, tr.tex);
      memcpy(&b[hd.off_tiles + t], &tr, 8);
    }
    if (mp.n_curves) {
      memcpy(&b[hd.off_curves], mp.curves, sizeof(double) * 8 * mp.n_curves);
      memcpy(&b[hd.off_heads], mp.curve_heads, sizeof(double) * 2 * mp.n_curves);
    }
    double* st = reinterpret_cast<double*>(&b[hd.off_static]);
    double* ob = reinterpret_cast<double*>(&b[hd.off_objs]);
    int si = 0, di = 0;
    RenderMapDev& rm = rmaps[mi];
    rm.grid_w = mp.grid_w; rm.grid_h = mp.grid_h; rm.n_obj = mp.n_objects; rm.n_tris = 0;
    for (int o = 0; o < mp.n_objects; ++o)
      if (mp.objects[o].mesh_id >= 0 && mp.objects[o].mesh_id < (int)h->h_meshes.size()) rm.n_tris += h->h_meshes[mp.objects[o].mesh_id].n_tris;
    rm.tile_size = (float)mp.tile_size; rm.inv_tile_size = (float)(1.0 / mp.tile_size);
    rm.tile_off = (int32_t)rtiles.size(); rm.obj_off = (int32_t)robjs.size();
    for (int t = 0; t < nt; ++t) {
      const bool present = mp.tile_kind[t] != DTSIM_TILE_EMPTY;
      const int tex = mp.tile_tex[t] < 0 ? 0xFF : mp.tile_tex[t];
      rtiles.push_back((uint32_t)tex | ((uint32_t)(mp.tile_angle[t] & 3) << 8) | ((present ? 1u : 0u) << 15) |
                       ((mp.tile_tex[t] >= 0 ? 1u : 0u) << 14));
      TileLds tr{};
      tr.flags = present ? 1u : 0u;
      if (present && mp.tile_tex[t] >= 0 && mp.tile_tex[t] < (int)h->h_tex.size()) {
        const TexDev& td = h->h_tex[mp.tile_tex[t]];
        if (tex_w == 0) { tex_w = td.w; tex_h = td.h; }
        if (td.w != tex_w || td.h != tex_h)
          return fail(DTSIM_E_LIMIT, "map %d tile %d: all tile textures must share one size (%dx%d vs %dx%d)", mi, t, td.w, td.h, tex_w, tex_h);
        const int ang = mp.tile_angle[t] & 3;
        const float TW = (float)td.w, TH = (float)td.h;
        tr.tex_off = (uint32_t)td.off;
        tr.flags |= 2u;
        // u = {1-fx, fz, fx, 1-fz}[ang], v = {fz, fx, 1-fz, 1-fx}[ang]; x = u*TW - 0.5, y = v*TH - 0.5
        const bool swp = (ang & 1) != 0, flip_u = (ang == 0 || ang == 3), flip_v = (ang == 2 || ang == 3);
        const float mu = flip_u ? -TW : TW, mv = flip_v ? -TH : TH;
        tr.mxx = swp ? 0.f : mu; tr.mxz = swp ? mu : 0.f; tr.ox = flip_u ? TW - 0.5f : -0.5f;
        tr.myx = swp ? mv : 0.f; tr.myz = swp ? 0.f : mv; tr.oy = flip_v ? TH - 0.5f : -0.5f;
      }
      trecs.push_back(tr);
    }
    for (int o = 0; o < mp.n_objects; ++o) {
      const dtsim_object& ob_ = mp.objects[o];
      if (ob_.mesh_id >= h->n_meshes) return fail(DTSIM_E_INVALID, "map %d object %d: mesh %d not loaded", mi, o, ob_.mesh_id);
      int slot = -1;
      if (ob_.dynamic) {
        slot = di++;
        DynInit& d = dyn[(size_t)mi * DTSIM_MAX_DYNAMIC + slot];
        d.cx = ob_.pos[0]; d.cz = ob_.pos[2];
        memcpy(d.corners, ob_.corners, sizeof d.corners);
        memcpy(d.norm, ob_.norm, sizeof d.norm);
        d.heading_x = std::cos(ob_.angle); d.heading_z = -std::sin(ob_.angle);  // collision.py:223-230
        d.angle = ob_.angle; d.safety_radius = ob_.safety_radius;
        d.walk_distance = ob_.walk_distance; d.vel = ob_.vel; d.wait_time = ob_.wait_time; d.wiggle = ob_.wiggle;
        d.obj_index = o; d.kind = ob_.dynamic;
      } else if (ob_.collidable) {
        double* r = st + STATIC_WORDS * si++;
        memcpy(r, ob_.corners, 8 * sizeof(double));
        memcpy(r + 8, ob_.norm, 4 * sizeof(double));
        r[12] = ob_.pos[0]; r[13] = ob_.pos[2]; r[14] = ob_.safety_radius;
      }
      ob[o * OBJ_WORDS + 0] = ob_.pos[0]; ob[o * OBJ_WORDS + 1] = ob_.pos[2];
      ob[o * OBJ_WORDS + 2] = ob_.spawn_clear;
      ob[o * OBJ_WORDS + 3] = (double)(slot >= 0 ? slot : (ob_.optional ? -2 : -1));   // -2: optional static object
      ob[o * OBJ_WORDS + 4] = (double)ob_.light_freq; ob[o * OBJ_WORDS + 5] = (double)(ob_.light_pattern & 1);
      if (ob_.light_freq < 0) return fail(DTSIM_E_INVALID, "map %d object %d: light_freq %d", mi, o, ob_.light_freq);
      ObjInstDev oi{};
      oi.x = (float)ob_.pos[0]; oi.y = (float)ob_.pos[1]; oi.z = (float)ob_.pos[2];
      oi.scale = (float)ob_.scale; oi.yrot_deg = (float)(ob_.angle * (180.0 / 3.141592653589793));
      oi.mesh_id = ob_.mesh_id; oi.dyn_slot = slot;
      oi.light_tris = ob_.light_freq > 0 ? ob_.light_tris : 0; oi.light_tex0 = ob_.light_tex[0]; oi.light_tex1 = ob_.light_tex[1];
      if (oi.light_tris > 0 && (oi.light_tex0 >= h->n_tex || oi.light_tex1 >= h->n_tex))
        return fail(DTSIM_E_INVALID, "map %d object %d: light texture not loaded", mi, o);
      robjs.push_back(oi);
    }
    h->map_n_dyn[mi] = n_dyn; h->map_n_obj[mi] = mp.n_objects;
  }
  // ---- quad-layout fast path tables: possible when every tile texture is one square power-of-two size
  std::vector<uint32_t> qblocks, qtiles;
  int qlog2 = -1;
  float q_per_m = 0.f;
  if ((h->cfg.flags & DTSIM_F_RENDER) && tex_w == tex_h && tex_w >= 2) {
    const int S = tex_w;
    qlog2 = 0; while ((1 << qlog2) < S) ++qlog2;
    std::vector<int> block_of((size_t)std::max(h->n_tex, 1) * 4, -1);
    // the two one-record blocks: off-grid (meta high half 1) and untextured (meta 0); then the S x S blocks
    const uint32_t special[8] = {0u, 0u, 0u, 1u << 16, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u};   // untextured: white vertex colour
    qblocks.assign(special, special + 8);
    const size_t block_bytes = (size_t)S * S * 16;
    // second dword of a table entry: the mask of the record's byte offset inside its block (S = 256, q8_rec256) / of the cell number (other sizes): 0 for the
    // two one-record blocks.  S = 256: the blocks start at multiples of 1 MB (the offset is OR-ed in), the first one holds the two special records only.
    const uint32_t cell_sel = (qlog2 == 8) ? 0xFFFFFu : (uint32_t)(S * S - 1);
    const uint32_t zero_sel = 0u;
    if (qlog2 == 8) qblocks.resize(block_bytes / 4, 0u);
    int n_blocks = 0;
    for (int mi = 0; mi < n_maps; ++mi) {
      const dtsim_map& mp = maps[mi];
      RenderMapDev& rm = rmaps[mi];
      rm.qt_off = (int32_t)(qtiles.size() / 2); rm.qt_pitch = mp.grid_w + 2 * DT_QRING;
      q_per_m = std::max(q_per_m, (float)((double)S / mp.tile_size));
      for (int j = -DT_QRING; j < mp.grid_h + DT_QRING; ++j)
        for (int i = -DT_QRING; i < mp.grid_w + DT_QRING; ++i) {
          uint32_t off = 0u, sel = zero_sel;        // record 0: off-grid
          if (i >= 0 && j >= 0 && i < mp.grid_w && j < mp.grid_h) {
            const int t = j * mp.grid_w + i;
            if (mp.tile_kind[t] != DTSIM_TILE_EMPTY) {
              const int tx = mp.tile_tex[t];
              if (tx < 0 || tx >= (int)h->h_tex.size()) off = 16u;   // record 1: present but untextured
              else {
                int& b = block_of[(size_t)tx * 4 + (mp.tile_angle[t] & 3)];
                if (b < 0) { b = n_blocks++; build_quad_block(qblocks, h->h_pool.data() + h->h_tex[tx].off, S, mp.tile_angle[t] & 3); }
                off = (qlog2 == 8) ? (uint32_t)((size_t)(b + 1) << 20) : (uint32_t)(32 + (size_t)b * block_bytes); sel = cell_sel;
              }
            }
          }
          qtiles.push_back(off); qtiles.push_back(sel);
        }
    }
    if (32 + (size_t)(n_blocks + 1) * block_bytes >= ((size_t)1 << 32)) qlog2 = -1;   // 32-bit block offsets
    for (int mi = 0; mi < n_maps; ++mi)                                           // quad coordinates below 32768 (render.hip, Q8_SNAP): else the generic raster
      if ((size_t)(std::max(maps[mi].grid_w, maps[mi].grid_h) + 2 * DT_QRING) * S >= 32768) qlog2 = -1;
  }
  M.total_words = (int32_t)blobs.size();
  if ((size_t)M.total_words * 8 > 60000)
    return fail(DTSIM_E_LIMIT, "map tables %zu B exceed the 60 KB LDS staging budget", (size_t)M.total_words * 8);
  if (trecs.size() > DTSIM_LDS_TILES)
    return fail(DTSIM_E_LIMIT, "%zu tiles over all maps exceed the %d LDS raster records", trecs.size(), DTSIM_LDS_TILES);
  void* olds[] = {h->d_blobs, h->d_dyn, h->d_rmaps, h->d_rtiles, h->d_robjs, h->d_tilerecs, h->d_qtex, h->d_qtiles};
  for (void* p : olds) if (p) (void)hipFree(p);
  h->d_blobs = nullptr; h->d_dyn = nullptr; h->d_rmaps = nullptr; h->d_rtiles = nullptr; h->d_robjs = nullptr;
  h->d_tilerecs = nullptr; h->d_qtex = nullptr; h->d_qtiles = nullptr;
  h->n_qtiles = 0; h->qlog2 = 0; h->q_per_m = 0.f; h->q3_rows = 0;
  if (qlog2 > 0 && !qtiles.empty()) {
    HIPCHK(hipMalloc(&h->d_qtex, qblocks.size() * 4));
    HIPCHK(hipMemcpy(h->d_qtex, qblocks.data(), qblocks.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_qtiles, qtiles.size() * 4));
    HIPCHK(hipMemcpy(h->d_qtiles, qtiles.data(), qtiles.size() * 4, hipMemcpyHostToDevice));
    h->n_qtiles = (int)qtiles.size() / 2; h->qlog2 = qlog2; h->q_per_m = q_per_m;
    // ---- k_raster_v3 (render_v3.inc) reads the same pool and table through its own LDS layout: S = 256, padded grids up
    // to 32 x 24 tiles, up to 4 maps
    int rows = 0, cols = 0;
    for (int mi = 0; mi < n_maps; ++mi) { rows = std::max(rows, maps[mi].grid_h + 2 * DT_QRING); cols = std::max(cols, maps[mi].grid_w + 2 * DT_QRING); }
    if (qlog2 == 8 && rows <= 24 && cols <= 32 && n_maps * 32 <= 128 && !h->raster_old) h->q3_rows = rows;
  }
  HIPCHK(hipMalloc(&h->d_tilerecs, std::max<size_t>(trecs.size(), 1) * sizeof(TileLds)));
  if (!trecs.empty()) HIPCHK(hipMemcpy(h->d_tilerecs, trecs.data(), trecs.size() * sizeof(TileLds), hipMemcpyHostToDevice));
  if (h->d_stris) { (void)hipFree(h->d_stris); h->d_stris = nullptr; }
  if (h->d_objbox) { (void)hipFree(h->d_objbox); h->d_objbox = nullptr; }
  if (h->d_objmask) { (void)hipFree(h->d_objmask); h->d_objmask = nullptr; }
  h->max_tris = 0;
  for (auto& rm : rmaps) h->max_tris = std::max(h->max_tris, rm.n_tris);
  if (h->max_tris > 0 && (h->cfg.flags & DTSIM_F_RENDER)) {
    HIPCHK(hipMalloc(&h->d_stris, (sizeof(ScreenTri) + 16) * (size_t)h->max_tris * h->N));   // + the 16-byte screen boxes behind the triangles
    HIPCHK(hipMalloc(&h->d_objbox, sizeof(ObjBox) * (size_t)h->N * DTSIM_MAX_OBJECTS));
    const size_t n_blk = dt_raster_tiles(h->cfg.cam_width, h->cfg.cam_height) * 4;
    HIPCHK(hipMalloc(&h->d_objmask, n_blk * 16 + (size_t)DTSIM_MAX_MAPS * DTSIM_MAX_OBJECTS * 8 + (size_t)h->N * n_blk * 8));
  }
  h->n_tilerecs = (int)trecs.size();
  h->tex_w = tex_w ? tex_w : 1; h->tex_h = tex_h ? tex_h : 1;
  HIPCHK(hipMalloc(&h->d_blobs, blobs.size() * 8));
  HIPCHK(hipMemcpy(h->d_blobs, blobs.data(), blobs.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&h->d_dyn, dyn.size() * sizeof(DynInit)));
  HIPCHK(hipMemcpy(h->d_dyn, dyn.data(), dyn.size() * sizeof(DynInit), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&h->d_rmaps, rmaps.size() * sizeof(RenderMapDev)));
  HIPCHK(hipMemcpy(h->d_rmaps, rmaps.data(), rmaps.size() * sizeof(RenderMapDev), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&h->d_rtiles, std::max<size_t>(rtiles.size(), 1) * 4));
  if (!rtiles.empty()) HIPCHK(hipMemcpy(h->d_rtiles, rtiles.data(), rtiles.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&h->d_robjs, std::max<size_t>(robjs.size(), 1) * sizeof(ObjInstDev)));
  if (!robjs.empty()) HIPCHK(hipMemcpy(h->d_robjs, robjs.data(), robjs.size() * sizeof(ObjInstDev), hipMemcpyHostToDevice));
  M.blobs = h->d_blobs;
  M.dyn = h->d_dyn;
  h->M = M;
  h->have_maps = true;
  // worlds must be re-created against the new maps
  HIPCHK(hipMemsetAsync(h->A.map_id, 0xFF, sizeof(int32_t) * (size_t)h->N, h->stream));
  h->have_reset = false;
  return DTSIM_OK;
}

int dtsim_set_distortion_lut(dtsim_t* h, const float* rmapx, const float* rmapy) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  h->leds_ok = false;
  h->render_tables = 0;           // the cached per-pixel / per-block render tables depend on this
  if (!h->d_lut) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  if ((rmapx == nullptr) != (rmapy == nullptr)) return fail(DTSIM_E_INVALID, "rmapx/rmapy must both be given");
  if (rmapx && !(h->cfg.flags & DTSIM_F_DISTORTION)) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_DISTORTION");
  HIPCHK(hipSetDevice(h->cfg.device));
  const int W = h->cfg.cam_width, H = h->cfg.cam_height;
  std::vector<float> lut((size_t)W * H * 4);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      long sx = c, sy = r;
      if (rmapx) {
        // cv2.remap(INTER_NEAREST): cvRound = round-half-to-even of the float map
        // (distortion.py:118-124); outside the source image => BORDER_CONSTANT 0.
        sx = std::lrint((double)rmapx[(size_t)r * W + c]);
        sy = std::lrint((double)rmapy[(size_t)r * W + c]);
      }
      float* o = &lut[((size_t)r * W + c) * 4];
      const bool ok = sx >= 0 && sx < W && sy >= 0 && sy < H;
      // NDC of the centre of rectilinear pixel (sy, sx); row 0 = image top (simulator.py:1949)
      o[0] = ok ? (float)((2.0 * (sx + 0.5)) / W - 1.0) : 0.f;
      o[1] = ok ? (float)(1.0 - (2.0 * (sy + 0.5)) / H) : 0.f;
      o[2] = ok ? 1.f : 0.f;
      o[3] = 0.f;
    }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice));
  h->have_lut = true;
  return DTSIM_OK;
}

static int check_states(const dtsim* h, const dtsim_init_state* st, int n, const uint8_t* mask) {
  for (int e = 0; e < n; ++e) {
    if (mask && !mask[e]) continue;
    const int mid = st[e].map_id & ~DTSIM_MAP_RELOAD;
    if (st[e].map_id < 0 || mid >= h->M.n_maps) return fail(DTSIM_E_INVALID, "state %d: map_id %d out of range", e, st[e].map_id);
  }
  return DTSIM_OK;
}

int dtsim_set_reset_sampler(dtsim_t* h, const dtsim_reset_sampler* sampler) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (!sampler) {
    if (h->d_sampler) { (void)hipFree(h->d_sampler); h->d_sampler = nullptr; }
    return DTSIM_OK;
  }
  if (!h->have_maps) return fail(DTSIM_E_STATE, "dtsim_set_reset_sampler before dtsim_set_maps");
  if (sampler->max_attempts <= 0 || !(sampler->accept_start_angle_deg > 0))
    return fail(DTSIM_E_INVALID, "sampler: max_attempts %d, accept_start_angle_deg %g", sampler->max_attempts, sampler->accept_start_angle_deg);
  for (int m = 0; m < h->M.n_maps; ++m) {
    const int i = sampler->start_tile[m][0], j = sampler->start_tile[m][1];
    if (i < 0) continue;
    if (i >= h->map_w[m] || j < 0 || j >= h->map_h[m]) return fail(DTSIM_E_INVALID, "sampler: start tile (%d,%d) outside map %d", i, j, m);
  }
  if (!h->d_sampler) HIPCHK(hipMalloc(&h->d_sampler, sizeof(dtsim_reset_sampler)));
  HIPCHK(hipMemcpy(h->d_sampler, sampler, sizeof(dtsim_reset_sampler), hipMemcpyHostToDevice));
  return DTSIM_OK;
}

int dtsim_reset_done(dtsim_t* h) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  if (!h->have_reset) return fail(DTSIM_E_STATE, "dtsim_reset_done before the first dtsim_reset");
  if (!h->d_sampler) return fail(DTSIM_E_STATE, "dtsim_reset_done needs dtsim_set_reset_sampler");
  h->rendered = false;
  HIPCHK(hipSetDevice(h->cfg.device));
  {
    ProfScope ps(h, DTSIM_KERNEL_RESET);
    dt_launch_reset(h->stream, h->A, h->M, step_params(h, 0), h->A.done, nullptr);   // mask = the done flags, on the device
  }
  HIPCHK(hipGetLastError());
  return DTSIM_OK;
}

int dtsim_reset(dtsim_t* h, const uint8_t* mask, const dtsim_init_state* states) {
  if (!h) return fail(DTSIM_E_INVALID, "null argument");
  if (!h->have_maps) return fail(DTSIM_E_STATE, "dtsim_reset before dtsim_set_maps");
  h->rendered = false;                               // the state moves on: the post-passes (dtsim_draw_lines / dtsim_draw_leds) need a new dtsim_render
  if (!states) {                                     // device-side sampling
    if (!h->d_sampler) return fail(DTSIM_E_STATE, "dtsim_reset(states = NULL) needs dtsim_set_reset_sampler");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (mask) HIPCHK(hipMemcpyAsync(h->d_mask, mask, (size_t)h->N, hipMemcpyHostToDevice, h->stream));
    {
      ProfScope ps(h, DTSIM_KERNEL_RESET);
      dt_launch_reset(h->stream, h->A, h->M, step_params(h, 0), mask ? h->d_mask : nullptr, nullptr);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_reset = true;
    return DTSIM_OK;
  }
  int rc = check_states(h, states, h->N, mask);
  if (rc) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipMemcpyAsync(h->d_states, states, sizeof(dtsim_init_state) * (size_t)h->N, hipMemcpyHostToDevice, h->stream));
  if (mask) HIPCHK(hipMemcpyAsync(h->d_mask, mask, (size_t)h->N, hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, DTSIM_KERNEL_RESET);
    dt_launch_reset(h->stream, h->A, h->M, step_params(h, 0), mask ? h->d_mask : nullptr, h->d_states);
  }
  HIPCHK(hipGetLastError());
  // the host buffers may be reused by the caller as soon as we return
  HIPCHK(hipStreamSynchronize(h->stream));
  if (!mask) h->have_reset = true;
  else h->have_reset = true;
  return DTSIM_OK;
}

int dtsim_set_spawn_pool(dtsim_t* h, const dtsim_init_state* pool, int n_pool) {
  if (!h || !pool || n_pool <= 0) return fail(DTSIM_E_INVALID, "bad pool");
  if (!h->have_maps) return fail(DTSIM_E_STATE, "dtsim_set_spawn_pool before dtsim_set_maps");
  int rc = check_states(h, pool, n_pool, nullptr);
  if (rc) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->d_pool) { (void)hipFree(h->d_pool); h->d_pool = nullptr; }
  HIPCHK(hipMalloc(&h->d_pool, sizeof(dtsim_init_state) * (size_t)n_pool));
  HIPCHK(hipMemcpy(h->d_pool, pool, sizeof(dtsim_init_state) * (size_t)n_pool, hipMemcpyHostToDevice));
  h->n_pool = n_pool;
  return DTSIM_OK;
}

int dtsim_step(dtsim_t* h, const void* actions, int n_steps, int actions_on_device) {
  return dtsim_step_ex(h, actions, n_steps, actions_on_device, 0u);
}

int dtsim_step_ex(dtsim_t* h, const void* actions, int n_steps, int actions_on_device, uint32_t flags) {
  if (!h || !actions || n_steps <= 0) return fail(DTSIM_E_INVALID, "bad argument");
  if (flags & ~(uint32_t)(DTSIM_STEP_ONE_UPDATE | DTSIM_STEP_POSE_ONLY)) return fail(DTSIM_E_INVALID, "unknown step flags 0x%x", flags);
  if (!h->have_maps || !h->have_reset) return fail(DTSIM_E_STATE, "dtsim_step before dtsim_set_maps/dtsim_reset");
  h->rendered = false;                               // the cameras / projected triangles of the last render pass describe the state before this step
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t esz = (h->cfg.flags & DTSIM_F_ACTIONS_F64) ? 8 : 4;
  const size_t bytes = (size_t)n_steps * h->N * 2 * esz;
  const void* dptr = actions;
  if (!actions_on_device) {
    if (bytes > h->actions_cap) {
      HIPCHK(hipStreamSynchronize(h->stream));
      if (h->d_actions) (void)hipFree(h->d_actions);
      h->d_actions = nullptr; h->actions_cap = 0;
      HIPCHK(hipMalloc(&h->d_actions, bytes));
      h->actions_cap = bytes;
    }
    // pageable host memory: hipMemcpyAsync stages synchronously, so the caller may
    // reuse `actions` on return
    HIPCHK(hipMemcpyAsync(h->d_actions, actions, bytes, hipMemcpyHostToDevice, h->stream));
    dptr = h->d_actions;
  }
  {
    ProfScope ps(h, DTSIM_KERNEL_STEP);
    StepParams sp = step_params(h, n_steps);
    sp.step_flags = flags;
    dt_launch_step(h->stream, h->A, h->M, sp, dptr, h->d_pool);
  }
  HIPCHK(hipGetLastError());
  return DTSIM_OK;
}

int dtsim_render(dtsim_t* h) { return dtsim_render_ex(h, 0u); }

int dtsim_render_ex(dtsim_t* h, uint32_t flags) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  if (flags & ~(uint32_t)(DTSIM_RENDER_SEGMENT | DTSIM_RENDER_GL_FILTER)) return fail(DTSIM_E_INVALID, "unknown render flags 0x%x", flags);
  const bool segment = (flags & DTSIM_RENDER_SEGMENT) != 0;
  if (segment && !h->d_texels_seg) return fail(DTSIM_E_STATE, "DTSIM_RENDER_SEGMENT before dtsim_set_segment_assets");
  if (!h->frames) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  if (!h->have_maps || !h->have_reset) return fail(DTSIM_E_STATE, "dtsim_render before dtsim_set_maps/dtsim_reset");
  if (!h->have_lut) return fail(DTSIM_E_STATE, "DTSIM_F_DISTORTION set but dtsim_set_distortion_lut was not called");
  HIPCHK(hipSetDevice(h->cfg.device));
  RenderParams R{};
  R.N = h->N; R.W = h->cfg.cam_width; R.H = h->cfg.cam_height;
  R.distortion = (h->cfg.flags & DTSIM_F_DISTORTION) ? 1 : 0;
  R.domain_rand = (h->cfg.flags & DTSIM_F_DOMAIN_RAND) ? 1 : 0;
  R.n_maps = h->M.n_maps;
#ifdef DT_RASTER_NO_MSAA   // profiling ablation, tools/build_variant.sh only: never in the product build
  R.no_msaa = 1;
#else
  R.no_msaa = 0;
#endif
  R.frames = h->frames; R.lut = h->d_lut; R.texels = segment ? h->d_texels_seg : h->d_texels; R.tex = h->d_tex;
  R.segment = segment ? 1 : 0; R.mesh_seg = h->d_mesh_seg;
  R.maps = h->d_rmaps; R.tiles = h->d_rtiles; R.objs = h->d_robjs; R.meshes = h->d_meshes; R.tris = h->d_tris;
  R.envcam = h->d_envcam;
  R.max_tris = h->d_stris ? h->max_tris : 0; R.stris = h->d_stris; R.objbox = h->d_objbox;
  R.tribox = h->d_stris ? reinterpret_cast<float4*>(h->d_stris + (size_t)h->max_tris * h->N) : nullptr;
  R.blockbox = reinterpret_cast<float*>(h->d_objmask);
  R.objrange = h->d_objmask ? reinterpret_cast<uint2*>(reinterpret_cast<char*>(h->d_objmask) + dt_raster_tiles(R.W, R.H) * 4 * 16) : nullptr;
  R.objmask = h->d_objmask ? reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(R.objrange) + (size_t)DTSIM_MAX_MAPS * DTSIM_MAX_OBJECTS * 8) : nullptr;
  R.pad4_ = 0;
  R.queue = h->d_queue; R.qcount = h->d_qcount;
  R.dbg = nullptr;
  const size_t n_wg_ = dt_raster_tiles(R.W, R.H) * (((size_t)h->N + DT_ENVS_PER_BLOCK - 1) / DT_ENVS_PER_BLOCK);
  R.work = h->d_qcount + n_wg_ * 4 + 8; R.items = h->d_items; R.items2 = h->d_items + n_wg_ * DT_ITEMS_PER_WG; R.qend = h->d_qend;
  if (getenv("DTSIM_DEBUG_QUEUE")) {
    R.dbg = h->d_qcount + n_wg_ * 4;
    HIPCHK(hipMemsetAsync(R.dbg, 0, 8 * sizeof(int32_t), h->stream));
  }
  R.tile_recs = h->d_tilerecs; R.n_tile_recs = h->n_tilerecs; R.tex_w = h->tex_w; R.tex_h = h->tex_h;
  R.qtex = (flags & DTSIM_RENDER_GL_FILTER) ? nullptr : h->d_qtex;   // no quad records: the generic raster (llvmpipe's GL_LINEAR arithmetic) takes the pass
  R.qtiles = h->d_qtiles; R.n_qtiles = h->n_qtiles; R.qlog2 = h->qlog2; R.q_per_m = h->q_per_m;
  R.pixtab = h->d_pixtab;
  R.q3_rows = h->q3_rows;
  R.envpos = reinterpret_cast<int32_t*>((char*)h->d_envcam + (size_t)h->N * (128 + 64 + 64));
  R.envv = (char*)h->d_envcam + (((size_t)h->N * (128 + 64 + 64 + 4) + 63) / 64) * 64;
  R.envd = (char*)R.envv + ((size_t)h->N + 1) * 64;
  R.dump = (char*)h->d_pixtab + (size_t)R.W * R.H * 64;
  R.qmax_tiles = 0;
  for (int mi = 0; mi < h->M.n_maps; ++mi) R.qmax_tiles = std::max(R.qmax_tiles, std::max(h->map_w[mi], h->map_h[mi]) + 2 * DT_QRING);
#ifdef DT_WAVE_SPANS   // experiment: [2][2048][4][8] spans of the exact-path kernels, then [raster workgroups][4 wavefronts][4] stamps of k_raster_v3
  static unsigned long long* d_spans = nullptr;
  const size_t n_spans = (size_t)2 * 2048 * 4 * 8 + dt_raster_tiles(R.W, R.H) * (((size_t)h->N + DT_ENVS_PER_BLOCK - 1) / DT_ENVS_PER_BLOCK) * 4 * 4;
  if (getenv("DTSIM_WAVE_SPANS") && !d_spans) HIPCHK(hipMalloc(&d_spans, n_spans * 8));
  R.spans = getenv("DTSIM_WAVE_SPANS") ? d_spans : nullptr;
  if (R.spans) HIPCHK(hipMemsetAsync(R.spans, 0, n_spans * 8, h->stream));
#endif
  {
    ProfScope ps(h, DTSIM_KERNEL_RENDER);
    h->render_tables = dt_launch_render(h->stream, h->A, R, h->render_tables, h->overlap.parts > 1 ? &h->overlap : nullptr);
  }
  HIPCHK(hipGetLastError());
  h->rendered = true; h->last_R = R; h->last_segment = segment; h->leds_ok = true;
#ifdef DT_WAVE_SPANS
  if (R.spans) {   // the spans of the last render -> the file DTSIM_WAVE_SPANS names (tools/wave_spans.py reads it)
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> sp(n_spans);
    HIPCHK(hipMemcpy(sp.data(), R.spans, sp.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(getenv("DTSIM_WAVE_SPANS"), "wb")) { fwrite(sp.data(), 8, sp.size(), f); fclose(f); }
  }
#endif
  if (getenv("DTSIM_DEBUG_QUEUE") || getenv("DTSIM_DEBUG_TIMERS")) {   // profiling aid: how many pixels took the exact MSAA path (DEBUG_TIMERS: without the in-kernel counters)
    HIPCHK(hipStreamSynchronize(h->stream));
    {  // phase timers of the DT_Q_TIMING build variant (zero otherwise)
      unsigned long long tc[4];
      char* dbgp = (char*)h->d_pixtab + (size_t)R.W * R.H * 64 + 1024;
      HIPCHK(hipMemcpy(tc, dbgp, sizeof tc, hipMemcpyDeviceToHost));
      if (tc[3]) fprintf(stderr, "[dtsim] k_raster_q phase cycles per wavefront iteration: issue %.0f, wait for quads %.0f, filter+slow+transpose %.0f  (%llu iterations)\n",
                         (double)tc[0] / tc[3], (double)tc[1] / tc[3], (double)tc[2] / tc[3], tc[3]);
      HIPCHK(hipMemset(dbgp, 0, sizeof tc));
      unsigned long long t3[8];                      // DT_V3_TIMING build variant: k_raster_v3 phase cycles (second KB of the scratch)
      HIPCHK(hipMemcpy(t3, dbgp + 512, sizeof t3, hipMemcpyDeviceToHost));
      if (t3[7]) fprintf(stderr, "[dtsim] k_raster_v3 cycles per wavefront iteration (wall clock of the wavefront): issue %.0f, store+prefetch %.0f, "
                         "wait for the records %.0f, weights+filter %.0f, transpose %.0f, slow+append %.0f; whole iteration %.0f  (%llu iterations)\n",
                         (double)t3[0] / t3[7], (double)t3[1] / t3[7], (double)t3[2] / t3[7], (double)t3[3] / t3[7], (double)t3[4] / t3[7],
                         (double)t3[5] / t3[7], (double)t3[6] / t3[7], t3[7]);
      HIPCHK(hipMemset(dbgp + 512, 0, sizeof t3));
      unsigned long long tr[13];                     // DT_RES_TIMING build variant: k_resolve phase cycles
      HIPCHK(hipMemcpy(tr, dbgp + 64, sizeof tr, hipMemcpyDeviceToHost));
      if (tr[8]) fprintf(stderr, "[dtsim] k_resolve cycles per wavefront: total %.0f = item setup %.0f + entry load %.0f + mesh stream %.0f + z-buffer %.0f + shade %.0f; "
                         "%.1f items, %.1f batches per wavefront (%llu wavefronts)\n", (double)tr[7] / tr[8], (double)tr[0] / tr[8], (double)tr[1] / tr[8], (double)tr[2] / tr[8],
                         (double)tr[3] / tr[8], (double)tr[4] / tr[8], (double)tr[6] / tr[8], (double)tr[5] / tr[8], tr[8]);
      if (tr[8]) fprintf(stderr, "[dtsim] k_resolve: longest wavefront %llu, longest batch %llu cycles; %llu (batch, env) pairs with objects in %llu batches, at most %llu in one batch\n",
                         tr[9], tr[10], tr[11], tr[5], tr[12]);
      HIPCHK(hipMemset(dbgp + 64, 0, sizeof tr));
      int32_t ro[12];                                // DT_RO_STATS build variant: k_resolve_obj's z-buffer
      HIPCHK(hipMemcpy(ro, dbgp + 768, sizeof ro, hipMemcpyDeviceToHost));
      unsigned long long rp; memcpy(&rp, ro + 6, 8);
      if (ro[4]) fprintf(stderr, "[dtsim] k_resolve_obj z-buffer: %d calls (%d triangle-parallel), %.1f staged triangles and %.1f pixels per call, "
                         "%.2f box candidates per pixel, %.2f passes per call (max over lanes), %.2f if the pairs were spread evenly\n",
                         ro[4], ro[5], (double)ro[2] / ro[4], (double)ro[3] / ro[4], ro[3] ? (double)ro[8] / ro[3] : 0.0, (double)ro[9] / ro[4], (double)ro[10] / ro[4]);
      HIPCHK(hipMemset(dbgp + 768, 0, sizeof ro));
    }
    const size_t npix = (size_t)R.W * R.H;
    const size_t n_wg = dt_raster_tiles(h->cfg.cam_width, h->cfg.cam_height) * (((size_t)h->N + DT_ENVS_PER_BLOCK - 1) / DT_ENVS_PER_BLOCK);
    std::vector<int32_t> qc(n_wg * 4);
    HIPCHK(hipMemcpy(qc.data(), h->d_qcount, qc.size() * 4, hipMemcpyDeviceToHost));
    long long tot = 0, mx = 0, iters = 0, nonempty = 0;
    for (int32_t v : qc) { tot += v; mx = std::max<long long>(mx, v); iters += (v + 63) / 64; nonempty += v > 0; }
    fprintf(stderr, "[dtsim] resolve: %lld of %zu wavefront regions non-empty, %lld 64-lane iterations, lane utilisation %.1f%%\n",
            nonempty, qc.size(), iters, iters ? 100.0 * tot / (64.0 * iters) : 0.0);
    int32_t dbg[8];
    HIPCHK(hipMemcpy(dbg, h->d_qcount + n_wg * 4, sizeof dbg, hipMemcpyDeviceToHost));
    unsigned long long pairs; memcpy(&pairs, dbg + 6, 8);
    fprintf(stderr, "[dtsim] resolve mesh pass: %d (batch,env) pairs, %d objects streamed, %d z-buffer calls (%d triangle-parallel), "
                    "%d triangles staged, %d pixels, %llu pixel x triangle tests\n", dbg[0], dbg[1], dbg[4], dbg[5], dbg[2], dbg[3], pairs);
    fprintf(stderr, "[dtsim] exact-path pixels: %lld of %zu (%.2f%%), max per wavefront region %lld\n", tot, npix * h->N,
            100.0 * tot / (double)(npix * h->N), mx);
  }
  return DTSIM_OK;
}

int dtsim_draw_lines(dtsim_t* h, const float* lines, const int32_t* env_idx, int n) {
  if (!h || (n > 0 && !lines)) return fail(DTSIM_E_INVALID, "null argument");
  if (n < 0) return fail(DTSIM_E_INVALID, "n = %d", n);
  if (!h->frames) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  if (!h->have_maps || !h->have_reset || !h->have_lut || !h->rendered) return fail(DTSIM_E_STATE, "dtsim_draw_lines before the first dtsim_render (the pass writes the cameras the lines go through)");
  if (n == 0) return DTSIM_OK;
  for (int i = 0; i < n; ++i) {
    const int e = env_idx ? env_idx[i] : 0;
    if (e < 0 || e >= h->N) return fail(DTSIM_E_INVALID, "env_idx[%d] = %d out of range [0, %d)", i, e, h->N);
    if (env_idx && i && env_idx[i] < env_idx[i - 1]) return fail(DTSIM_E_INVALID, "env_idx must be non-decreasing (segments grouped by env)");
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->lines_cap < n) {
    if (h->d_lines) { HIPCHK(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_lines); h->d_lines = nullptr; }
    h->lines_cap = std::max(n, 1024);
    HIPCHK(hipMalloc(&h->d_lines, sizeof(float) * 9 * (size_t)h->lines_cap));
  }
  HIPCHK(hipMemcpyAsync(h->d_lines, lines, sizeof(float) * 9 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  RenderParams R{};
  R.N = h->N; R.W = h->cfg.cam_width; R.H = h->cfg.cam_height; R.frames = h->frames; R.lut = h->d_lut; R.envcam = h->d_envcam;
  int i0 = 0;
  while (i0 < n) {                                    // one launch per env that has segments
    const int e = env_idx ? env_idx[i0] : 0;
    int i1 = i0;
    while (i1 < n && (env_idx ? env_idx[i1] : 0) == e) ++i1;
    dt_launch_overlay_lines(h->stream, R, h->d_lines, i0, i1 - i0, e);
    i0 = i1;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));           // `lines` is pageable host memory: the copy must have left it
  return DTSIM_OK;
}

int dtsim_draw_leds(dtsim_t* h, const float* spheres, const int32_t* env_idx, int n) {
  if (!h || (n > 0 && !spheres)) return fail(DTSIM_E_INVALID, "null argument");
  if (n < 0) return fail(DTSIM_E_INVALID, "n = %d", n);
  if (!h->frames) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  if (!h->rendered || !h->leds_ok || h->last_segment) return fail(DTSIM_E_STATE, "dtsim_draw_leds needs a preceding dtsim_render (colour view): it tests the spheres against that pass's scene");
  if (n == 0) return DTSIM_OK;
  for (int i = 0; i < n; ++i) {
    const int e = env_idx ? env_idx[i] : 0;
    if (e < 0 || e >= h->N) return fail(DTSIM_E_INVALID, "env_idx[%d] = %d out of range [0, %d)", i, e, h->N);
    if (env_idx && i && env_idx[i] < env_idx[i - 1]) return fail(DTSIM_E_INVALID, "env_idx must be non-decreasing (spheres grouped by env, in draw order)");
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->leds_cap < n) {
    if (h->d_leds) { HIPCHK(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_leds); h->d_leds = nullptr; }
    h->leds_cap = std::max(n, 256);
    HIPCHK(hipMalloc(&h->d_leds, sizeof(float) * 8 * (size_t)h->leds_cap));
  }
  HIPCHK(hipMemcpyAsync(h->d_leds, spheres, sizeof(float) * 8 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  RenderParams R = h->last_R;
  R.frames = h->frames;                               // (dtsim_bind_frames may have moved the output since the pass)
  int i0 = 0;
  while (i0 < n) {                                    // one launch per env that has spheres
    const int e = env_idx ? env_idx[i0] : 0;
    int i1 = i0;
    while (i1 < n && (env_idx ? env_idx[i1] : 0) == e) ++i1;
    dt_launch_overlay_leds(h->stream, R, h->d_leds, i0, i1 - i0, e);
    i0 = i1;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));           // `spheres` is pageable host memory: the copy must have left it
  return DTSIM_OK;
}

void* dtsim_frames_devptr(dtsim_t* h) { return h ? h->frames : nullptr; }
size_t dtsim_frames_bytes(const dtsim_t* h) { return h ? h->frames_bytes : 0; }

int dtsim_bind_frames(dtsim_t* h, void* devptr) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  if (!h->frames_own) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  h->frames = devptr ? (uint8_t*)devptr : h->frames_own;
  return DTSIM_OK;
}

// ---- RCCL all-gather of the frame batch (include/dtsim.h) ---------------------------------
// librccl is resolved at first use: the product path of a single GPU never loads it.
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
nccl_allgather_fn g_nccl_allgather = nullptr;
nccl_errstr_fn g_nccl_errstr = nullptr;
std::once_flag g_nccl_once;
std::string g_nccl_why;                               // why the library / symbol could not be had (kept: dlerror() is one-shot)
void resolve_rccl() {
  std::call_once(g_nccl_once, [] {
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);    // (the process's own copy when torch already loaded it)
      if (lib) break;
      if (const char* e = dlerror()) g_nccl_why = e;
    }
    if (!lib) return;
    g_nccl_allgather = reinterpret_cast<nccl_allgather_fn>(dlsym(lib, "ncclAllGather"));
    g_nccl_errstr = reinterpret_cast<nccl_errstr_fn>(dlsym(lib, "ncclGetErrorString"));
    if (!g_nccl_allgather) { const char* e = dlerror(); g_nccl_why = e ? e : "ncclAllGather not exported"; }
  });
}
}  // namespace

int dtsim_allgather_frames(dtsim_t* h, void* nccl_comm, void* recv, const void* send, size_t send_bytes) {
  if (!h || !nccl_comm || !recv) return fail(DTSIM_E_INVALID, "bad argument");
  if (!send) {
    if (!h->frames) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
    send = h->frames;
    send_bytes = dtsim_frames_bytes(h);
  }
  if (send_bytes == 0) return fail(DTSIM_E_INVALID, "empty send buffer");
  resolve_rccl();
  if (!g_nccl_allgather)
    return fail(DTSIM_E_STATE, "librccl.so (ncclAllGather) could not be loaded: %s", g_nccl_why.empty() ? "library or symbol not found" : g_nccl_why.c_str());
  HIPCHK(hipSetDevice(h->cfg.device));
  const int rc = g_nccl_allgather(send, recv, send_bytes, /*ncclUint8*/ 1, nccl_comm, h->stream);
  if (rc != 0) return fail(DTSIM_E_HIP, "ncclAllGather: %s", g_nccl_errstr ? g_nccl_errstr(rc) : "error");
  return DTSIM_OK;
}

int dtsim_observe(dtsim_t* h, void* out, int out_h, int out_w, int flags,
                  const int32_t* bounds_x, const int32_t* taps_x, int ksize_x,
                  const int32_t* bounds_y, const int32_t* taps_y, int ksize_y) {
  if (!h || !out) return fail(DTSIM_E_INVALID, "bad argument");
  if (!h->frames) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  const int W = h->cfg.cam_width, H = h->cfg.cam_height;
  if (out_h <= 0 || out_w <= 0) return fail(DTSIM_E_INVALID, "output size %dx%d", out_w, out_h);
  if ((out_w != W && (!bounds_x || !taps_x || ksize_x <= 0)) || (out_h != H && (!bounds_y || !taps_y || ksize_y <= 0)))
    return fail(DTSIM_E_INVALID, "resampling tables missing for a resized axis");
  HIPCHK(hipSetDevice(h->cfg.device));
  // tables: cached per (out_h, out_w); [bx | kkx | by | kky] in one device buffer
  if (h->obs_h != out_h || h->obs_w != out_w || !h->d_obs_tab) {
    std::vector<int32_t> tab;
    std::vector<int32_t> by(2 * (size_t)out_h);
    if (out_w != W) { tab.insert(tab.end(), bounds_x, bounds_x + 2 * (size_t)out_w); tab.insert(tab.end(), taps_x, taps_x + (size_t)out_w * ksize_x); }
    else ksize_x = 0;
    h->obs_off_by = tab.size();
    if (out_h != H) for (int i = 0; i < 2 * out_h; ++i) by[i] = bounds_y[i];
    else { for (int i = 0; i < out_h; ++i) { by[2 * i] = i; by[2 * i + 1] = 1; } ksize_y = 0; }
    for (int i = 0; i < out_h; ++i)
      if (by[2 * i] < 0 || by[2 * i + 1] <= 0 || by[2 * i] + by[2 * i + 1] > H || (i && by[2 * i] < by[2 * i - 2]))
        return fail(DTSIM_E_INVALID, "bounds_y[%d] = (%d, %d) out of range / not monotone", i, by[2 * i], by[2 * i + 1]);
    if (out_w != W)
      for (int i = 0; i < out_w; ++i)
        if (bounds_x[2 * i] < 0 || bounds_x[2 * i + 1] <= 0 || bounds_x[2 * i + 1] > ksize_x || bounds_x[2 * i] + bounds_x[2 * i + 1] > W)
          return fail(DTSIM_E_INVALID, "bounds_x[%d] = (%d, %d) out of range", i, bounds_x[2 * i], bounds_x[2 * i + 1]);
    tab.insert(tab.end(), by.begin(), by.end());
    if (out_h != H) {
      for (int i = 0; i < out_h; ++i) if (by[2 * i + 1] > ksize_y) return fail(DTSIM_E_INVALID, "bounds_y[%d] count > ksize_y", i);
      tab.insert(tab.end(), taps_y, taps_y + (size_t)out_h * ksize_y);
    }
    // rows per workgroup: as many output rows as keep the uint8 intermediate (+ staging) within 48 KB of LDS
    const size_t stage = DT_OBS_STAGE_ROWS * (((size_t)W * 3 + 3) / 4) * 4 + 32, tabs = (out_w != W && ksize_x <= 9) ? (size_t)out_w * 11 * 4 : 0;
    const size_t budget = DT_OBS_LDS_KB * 1024 - stage - tabs - 32;
    const int max_rows = (int)std::min<size_t>((size_t)H, budget / ((size_t)out_w * 3));
    int rpb = 0, need = 0;
    for (int cand = 1; cand <= out_h; ++cand) {
      int worst = 0;
      for (int o0 = 0; o0 < out_h; o0 += cand) {
        const int o1 = std::min(o0 + cand, out_h) - 1;
        worst = std::max(worst, by[2 * o1] + by[2 * o1 + 1] - by[2 * o0]);
      }
      if (worst > max_rows) break;
      rpb = cand; need = worst;
      if (cand >= DT_OBS_MAX_RPB) break;            // enough rows per workgroup; keep the grid large
    }
    if (rpb == 0) return fail(DTSIM_E_LIMIT, "observation %dx%d: one output row needs more input rows than fit in LDS", out_w, out_h);
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_obs_tab) { (void)hipFree(h->d_obs_tab); h->d_obs_tab = nullptr; }
    HIPCHK(hipMalloc(&h->d_obs_tab, tab.size() * sizeof(int32_t)));
    HIPCHK(hipMemcpy(h->d_obs_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->obs_h = out_h; h->obs_w = out_w; h->obs_kx = ksize_x; h->obs_ky = ksize_y; h->obs_rpb = rpb; h->obs_rows_in = need;
    // power-of-two down-scaling: interior columns / rows with identical small-integer taps (k_observe's dot4 / two-lane paths)
    h->obs_fast = ObserveParams{};
    auto uniform = [&](const int32_t* bounds, const int32_t* taps, int ksize, int n_in, int n_out, int S, uint32_t* w, int* sh) -> bool {
      if (n_out < 3 || n_out * S != n_in || 2 * S > ksize || 2 * S > 16) return false;
      const int32_t* k1 = taps + (size_t)1 * ksize;
      int common = 22;                                  // trailing zero bits shared by the taps of column 1
      for (int t = 0; t < 2 * S; ++t) { if (k1[t] <= 0) return false; common = std::min(common, __builtin_ctz((unsigned)k1[t])); }
      long long sum = 0;
      for (int t = 0; t < 2 * S; ++t) { const int q = k1[t] >> common; if (q > 255) return false; w[t] = (uint32_t)q; sum += q; }
      if (sum != (1ll << (22 - common)) || 22 - common < 1 || 22 - common > 7) return false;   // two-lane sums must stay below 2^16
      for (int o = 1; o < n_out - 1; ++o) {
        if (bounds[2 * o] != S * o - S / 2 || bounds[2 * o + 1] != 2 * S) return false;
        for (int t = 0; t < 2 * S; ++t) if (taps[(size_t)o * ksize + t] != k1[t]) return false;
      }
      *sh = 22 - common;
      return true;
    };
    if (out_w != W && ((size_t)W * 3) % 4 == 0) {
      for (int S : {4, 8}) {
        uint32_t w[16]; int sh = 0;
        if (!uniform(bounds_x, taps_x, ksize_x, W, out_w, S, w, &sh)) continue;
        ObserveParams& F = h->obs_fast;
        F.hfast = S; F.hsh = sh;
        const int start = -3 * S / 2;                  // first byte of a column's window relative to 3 S ox
        F.hoff = start & ~3;                           // (two's complement: rounds towards minus infinity)
        F.hn = (3 * 2 * S + (start - F.hoff) + 3) / 4;
        if (F.hn != 7 && F.hn != 12) { F.hfast = 0; continue; }
        for (int b = 0; b < 3 * 2 * S; ++b) {
          const int p = b + (start - F.hoff);
          F.hw[b % 3][p / 4] |= w[b / 3] << (8 * (p % 4));
        }
        break;
      }
    }
    if (out_h != H && ((size_t)out_w * 3) % 4 == 0) {
      for (int S : {2, 4, 8}) {
        uint32_t w[16]; int sh = 0;
        if (!uniform(by.data(), taps_y, ksize_y, H, out_h, S, w, &sh)) continue;
        h->obs_fast.vfast = S; h->obs_fast.vsh = sh;
        for (int t = 0; t < 2 * S; ++t) h->obs_fast.vw[t] = w[t];
        break;
      }
    }
    if (getenv("DTSIM_OBSERVE_GENERIC")) h->obs_fast = ObserveParams{};   // A/B switch: the table-driven paths only
  }
  ObserveParams P{};
  P.N = h->N; P.H = H; P.W = W; P.oh = out_h; P.ow = out_w; P.kx = h->obs_kx; P.ky = h->obs_ky;
  P.rows_per_block = h->obs_rpb; P.max_rows_in = h->obs_rows_in;
  P.chw = (flags & DTSIM_OBS_CHW) ? 1 : 0; P.f32 = (flags & DTSIM_OBS_F32) ? 1 : 0;
  P.frames = h->frames; P.out = out;
  P.bx = h->d_obs_tab; P.kkx = h->d_obs_tab + (out_w != W ? 2 * (size_t)out_w : 0);
  P.by = h->d_obs_tab + h->obs_off_by; P.kky = P.by + 2 * (size_t)out_h;
  P.hfast = h->obs_fast.hfast; P.hn = h->obs_fast.hn; P.hoff = h->obs_fast.hoff; P.hsh = h->obs_fast.hsh;
  P.vfast = h->obs_fast.vfast; P.vsh = h->obs_fast.vsh;
  memcpy(P.hw, h->obs_fast.hw, sizeof P.hw); memcpy(P.vw, h->obs_fast.vw, sizeof P.vw);
  {
    ProfScope ps(h, DTSIM_KERNEL_OBSERVE);
    dt_launch_observe(h->stream, P);
  }
  HIPCHK(hipGetLastError());
  return DTSIM_OK;
}

int dtsim_observe_cubic(dtsim_t* h, void* out, int out_h, int out_w, int flags,
                        const int32_t* first_x, const int32_t* taps_x, const int32_t* first_y, const int32_t* taps_y) {
  if (!h || !out || !first_x || !taps_x || !first_y || !taps_y) return fail(DTSIM_E_INVALID, "bad argument");
  if (!h->frames) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_RENDER");
  const int W = h->cfg.cam_width, H = h->cfg.cam_height;
  if (out_h <= 0 || out_w <= 0) return fail(DTSIM_E_INVALID, "output size %dx%d", out_w, out_h);
  if ((size_t)W * 3 * sizeof(int32_t) + 16 > 64 * 1024) return fail(DTSIM_E_LIMIT, "frame rows of %d pixels do not fit the kernel's LDS row", W);
  for (int i = 0; i < out_w; ++i)
    if (first_x[i] < -3 || first_x[i] >= W) return fail(DTSIM_E_INVALID, "first_x[%d] = %d out of range", i, first_x[i]);
  for (int i = 0; i < out_h; ++i)
    if (first_y[i] < -3 || first_y[i] >= H) return fail(DTSIM_E_INVALID, "first_y[%d] = %d out of range", i, first_y[i]);
  for (size_t i = 0; i < 4 * (size_t)out_w; ++i) if (taps_x[i] < -32768 || taps_x[i] > 32767) return fail(DTSIM_E_INVALID, "taps_x[%zu] is not a 16-bit tap", i);
  for (size_t i = 0; i < 4 * (size_t)out_h; ++i) if (taps_y[i] < -32768 || taps_y[i] > 32767) return fail(DTSIM_E_INVALID, "taps_y[%zu] is not a 16-bit tap", i);
  HIPCHK(hipSetDevice(h->cfg.device));
  // tables: [first_x | taps_x | first_y | taps_y] in one device buffer, re-sent when they differ from the cached ones
  std::vector<int32_t> tab;
  tab.insert(tab.end(), first_x, first_x + out_w); tab.insert(tab.end(), taps_x, taps_x + 4 * (size_t)out_w);
  tab.insert(tab.end(), first_y, first_y + out_h); tab.insert(tab.end(), taps_y, taps_y + 4 * (size_t)out_h);
  if (tab != h->obsc_tab || !h->d_obsc_tab) {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_obsc_tab) { (void)hipFree(h->d_obsc_tab); h->d_obsc_tab = nullptr; }
    HIPCHK(hipMalloc(&h->d_obsc_tab, tab.size() * sizeof(int32_t)));
    HIPCHK(hipMemcpy(h->d_obsc_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->obsc_tab = tab;
  }
  ObserveParams P{};
  P.N = h->N; P.H = H; P.W = W; P.oh = out_h; P.ow = out_w; P.kx = 4; P.ky = 4;
  P.chw = (flags & DTSIM_OBS_CHW) ? 1 : 0; P.f32 = (flags & DTSIM_OBS_F32) ? 1 : 0;
  P.frames = h->frames; P.out = out;
  P.bx = h->d_obsc_tab; P.kkx = P.bx + out_w; P.by = P.kkx + 4 * (size_t)out_w; P.kky = P.by + out_h;
  {
    ProfScope ps(h, DTSIM_KERNEL_OBSERVE);
    dt_launch_observe_cubic(h->stream, P);
  }
  HIPCHK(hipGetLastError());
  return DTSIM_OK;
}

int dtsim_query(dtsim_t* h, int n, const int32_t* env_idx, const double* poses, double safety_factor,
                dtsim_probe* out) {
  if (!h || n <= 0 || !env_idx || !poses || !out) return fail(DTSIM_E_INVALID, "bad argument");
  if (!h->have_maps || !h->have_reset) return fail(DTSIM_E_STATE, "dtsim_query before dtsim_set_maps/dtsim_reset");
  for (int i = 0; i < n; ++i)
    if (env_idx[i] < 0 || env_idx[i] >= h->N) return fail(DTSIM_E_INVALID, "env_idx[%d]=%d out of range", i, env_idx[i]);
  HIPCHK(hipSetDevice(h->cfg.device));
  if (n > h->q_cap) {
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_qenv) (void)hipFree(h->d_qenv);
    if (h->d_qpose) (void)hipFree(h->d_qpose);
    if (h->d_qout) (void)hipFree(h->d_qout);
    h->d_qenv = nullptr; h->d_qpose = nullptr; h->d_qout = nullptr; h->q_cap = 0;
    const int cap = n < 256 ? 256 : n;
    HIPCHK(hipMalloc(&h->d_qenv, sizeof(int32_t) * cap));
    HIPCHK(hipMalloc(&h->d_qpose, sizeof(double) * 3 * cap));
    HIPCHK(hipMalloc(&h->d_qout, sizeof(dtsim_probe) * cap));
    h->q_cap = cap;
  }
  HIPCHK(hipMemcpyAsync(h->d_qenv, env_idx, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_qpose, poses, sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, DTSIM_KERNEL_QUERY);
    dt_launch_query(h->stream, h->A, h->M, step_params(h, 0), n, h->d_qenv, h->d_qpose, safety_factor, h->d_qout);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, h->d_qout, sizeof(dtsim_probe) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DTSIM_OK;
}

// ---- field access -------------------------------------------------------------
namespace {
struct FieldDesc { void* base; size_t elem; int comps; size_t comp_stride_elems; bool planar; };

// planar == true: device layout is [comps][N]; the public layout is [N][comps].
bool field_desc(dtsim* h, int field, FieldDesc& d) {
  const SimArrays& A = h->A;
  const size_t N = (size_t)h->N;
  switch (field) {
    case DTSIM_FIELD_ANGLE: d = {A.angle, 8, 1, N, true}; return true;
    case DTSIM_FIELD_REWARD: d = {A.reward, 8, 1, N, true}; return true;
    case DTSIM_FIELD_DONE: d = {A.done, 1, 1, N, true}; return true;
    case DTSIM_FIELD_DONE_CODE: d = {A.done_code, 1, 1, N, true}; return true;
    case DTSIM_FIELD_STEP_COUNT: d = {A.step_count, 4, 1, N, true}; return true;
    case DTSIM_FIELD_LANE: d = {A.lane, 8, 4, N, true}; return true;
    case DTSIM_FIELD_IN_LANE: d = {A.in_lane, 1, 1, N, true}; return true;
    case DTSIM_FIELD_PROX: d = {A.prox, 8, 1, N, true}; return true;
    case DTSIM_FIELD_SPEED: d = {A.speed, 8, 1, N, true}; return true;
    case DTSIM_FIELD_TIMESTAMP: d = {A.timestamp, 8, 1, N, true}; return true;
    case DTSIM_FIELD_WHEELS: d = {A.wheels, 8, 2, N, true}; return true;
    case DTSIM_FIELD_MAP_ID: d = {A.map_id, 4, 1, N, true}; return true;
    case DTSIM_FIELD_OBJ_ACTIVE: d = {A.ob_active, 1, DTSIM_MAX_DYNAMIC, N, true}; return true;
    case DTSIM_FIELD_OBJ_YROT: d = {A.ob_yrot, 8, DTSIM_MAX_DYNAMIC, N, true}; return true;
    case DTSIM_FIELD_OBJ_VISIBLE: d = {A.ob_visible, 1, DTSIM_MAX_OBJECTS, N, true}; return true;
    case DTSIM_FIELD_OBJ_LIGHT: d = {A.ob_light, 1, DTSIM_MAX_OBJECTS, N, true}; return true;
    case DTSIM_FIELD_OBJ_Y: d = {A.ob_cy, 8, DTSIM_MAX_DYNAMIC, N, true}; return true;
    case DTSIM_FIELD_EPISODE: d = {A.episode, 4, 1, N, true}; return true;
    case DTSIM_FIELD_CAMERA: d = {A.cam, 4, 6, N, true}; return true;
    case DTSIM_FIELD_COLORS: d = {A.colors, 4, 16, N, true}; return true;
    case DTSIM_FIELD_WHEEL_DIST: d = {A.wheel_dist, 8, 1, N, true}; return true;
    default: return false;
  }
}

size_t public_bytes(const dtsim* h, int field) {
  const size_t N = (size_t)h->N;
  switch (field) {
    case DTSIM_FIELD_POS: return N * 3 * 8;
    case DTSIM_FIELD_TILE: return N * 2 * 4;
    case DTSIM_FIELD_OBJ_CENTER: return N * DTSIM_MAX_DYNAMIC * 2 * 8;
    case DTSIM_FIELD_OBJ_PARAMS: return N * DTSIM_MAX_DYNAMIC * 3 * 8;
    case DTSIM_FIELD_OBJ_EXTRA: return N * DTSIM_MAX_DYNAMIC * 5 * 8;
    case DTSIM_FIELD_STATE_BLOB: return h->slab_bytes;
    case DTSIM_FIELD_RENDER_POS: return N * 4;
    default: {
      FieldDesc d;
      if (!field_desc(const_cast<dtsim*>(h), field, d)) return 0;
      return N * d.comps * d.elem;
    }
  }
}

// copy `ncomp` planar device arrays ([c][N], given per-component base pointers) into the
// public [N][ncomp] layout (or back).
int xfer_planar(dtsim* h, void* const* bases, int ncomp, size_t elem, void* host, bool to_host) {
  const size_t N = (size_t)h->N;
  std::vector<char> tmp(N * elem);
  for (int c = 0; c < ncomp; ++c) {
    if (to_host) {
      if (bases[c] == nullptr) { memset(tmp.data(), 0, tmp.size()); }
      else {
        hipError_t e = hipMemcpy(tmp.data(), bases[c], N * elem, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(DTSIM_E_HIP, "hipMemcpy D2H: %s", hipGetErrorString(e));
      }
      for (size_t i = 0; i < N; ++i) memcpy((char*)host + (i * ncomp + c) * elem, tmp.data() + i * elem, elem);
    } else {
      if (bases[c] == nullptr) continue;
      for (size_t i = 0; i < N; ++i) memcpy(tmp.data() + i * elem, (const char*)host + (i * ncomp + c) * elem, elem);
      hipError_t e = hipMemcpy(bases[c], tmp.data(), N * elem, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(DTSIM_E_HIP, "hipMemcpy H2D: %s", hipGetErrorString(e));
    }
  }
  return DTSIM_OK;
}

__global__ void k_agent_info(SimArrays A, int e, dtsim_agent_info* out) {
  const size_t N = A.N;
  dtsim_agent_info r;
  r.pos[0] = A.pos_x[e]; r.pos[1] = 0.0; r.pos[2] = A.pos_z[e];
  r.angle = A.angle[e]; r.speed = A.speed[e]; r.timestamp = A.timestamp[e];
  r.wheels[0] = A.wheels[e]; r.wheels[1] = A.wheels[N + e];
  for (int k = 0; k < 4; ++k) r.lane[k] = A.lane[(size_t)k * N + e];
  r.prox = A.prox[e]; r.reward = A.reward[e];
  r.tile[0] = A.tile_i[e]; r.tile[1] = A.tile_j[e]; r.step_count = A.step_count[e];
  r.in_lane = A.in_lane[e]; r.done = A.done[e]; r.done_code = A.done_code[e]; r.pad = 0;
  *out = r;
}

int field_xfer(dtsim* h, int field, void* host, size_t bytes, bool to_host) {
  if (!h || !host) return fail(DTSIM_E_INVALID, "null argument");
  const size_t need = public_bytes(h, field);
  if (need == 0) return fail(DTSIM_E_INVALID, "unknown field %d", field);
  if (bytes != need) return fail(DTSIM_E_INVALID, "field %d: %zu bytes given, %zu expected", field, bytes, need);
  hipError_t e0 = hipSetDevice(h->cfg.device);
  if (e0 == hipSuccess) e0 = hipStreamSynchronize(h->stream);
  if (e0 != hipSuccess) return fail(DTSIM_E_HIP, "sync: %s", hipGetErrorString(e0));
  const SimArrays& A = h->A;
  const size_t N = (size_t)h->N;
  std::vector<void*> bases;
  switch (field) {
    case DTSIM_FIELD_STATE_BLOB: {
      hipError_t e = to_host ? hipMemcpy(host, h->slab, need, hipMemcpyDeviceToHost)
                             : hipMemcpy(h->slab, host, need, hipMemcpyHostToDevice);
      if (e != hipSuccess) return fail(DTSIM_E_HIP, "hipMemcpy blob: %s", hipGetErrorString(e));
      return DTSIM_OK;
    }
    case DTSIM_FIELD_RENDER_POS: {                    // read-only; the identity until a render pass ran in k_env_sort's order
      if (!to_host) return fail(DTSIM_E_INVALID, "DTSIM_FIELD_RENDER_POS is read-only");
      int32_t* out = static_cast<int32_t*>(host);
      if (h->render_tables & 4) {
        const int32_t* envpos = reinterpret_cast<const int32_t*>((const char*)h->d_envcam + N * (128 + 64 + 64));
        hipError_t e = hipMemcpy(out, envpos, N * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(DTSIM_E_HIP, "hipMemcpy render order: %s", hipGetErrorString(e));
      } else {
        for (size_t i = 0; i < N; ++i) out[i] = (int32_t)i;
      }
      return DTSIM_OK;
    }
    case DTSIM_FIELD_POS: bases = {A.pos_x, nullptr, A.pos_z}; return xfer_planar(h, bases.data(), 3, 8, host, to_host);
    case DTSIM_FIELD_TILE: bases = {A.tile_i, A.tile_j}; return xfer_planar(h, bases.data(), 2, 4, host, to_host);
    case DTSIM_FIELD_OBJ_CENTER:
      for (int d = 0; d < DTSIM_MAX_DYNAMIC; ++d) { bases.push_back(A.ob_cx + d * N); bases.push_back(A.ob_cz + d * N); }
      return xfer_planar(h, bases.data(), DTSIM_MAX_DYNAMIC * 2, 8, host, to_host);
    case DTSIM_FIELD_OBJ_PARAMS:
      for (int d = 0; d < DTSIM_MAX_DYNAMIC; ++d) { bases.push_back(A.ob_vel + d * N); bases.push_back(A.ob_wait + d * N); bases.push_back(A.ob_wiggle + d * N); }
      return xfer_planar(h, bases.data(), DTSIM_MAX_DYNAMIC * 3, 8, host, to_host);
    case DTSIM_FIELD_OBJ_EXTRA:
      for (int d = 0; d < DTSIM_MAX_DYNAMIC; ++d)
        for (int k = 0; k < 5; ++k) bases.push_back(A.ob_ext + ((size_t)k * DTSIM_MAX_DYNAMIC + d) * N);
      return xfer_planar(h, bases.data(), DTSIM_MAX_DYNAMIC * 5, 8, host, to_host);
    default: {
      FieldDesc d;
      field_desc(h, field, d);
      for (int c = 0; c < d.comps; ++c) bases.push_back((char*)d.base + c * d.comp_stride_elems * d.elem);
      return xfer_planar(h, bases.data(), d.comps, d.elem, host, to_host);
    }
  }
}
}  // namespace

int dtsim_read_agent(dtsim_t* h, int env, dtsim_agent_info* out) {
  if (!h || !out) return fail(DTSIM_E_INVALID, "null argument");
  if (env < 0 || env >= h->N) return fail(DTSIM_E_INVALID, "env %d out of range [0, %d)", env, h->N);
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->d_agent) HIPCHK(hipMalloc(&h->d_agent, sizeof(dtsim_agent_info)));
  hipLaunchKernelGGL(k_agent_info, dim3(1), dim3(1), 0, h->stream, h->A, env, h->d_agent);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, h->d_agent, sizeof(dtsim_agent_info), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DTSIM_OK;
}

int dtsim_read(dtsim_t* h, int field, void* dst, size_t bytes) { return field_xfer(h, field, dst, bytes, true); }
int dtsim_write(dtsim_t* h, int field, const void* src, size_t bytes) {
  if (h) h->rendered = false;                        // (any field may move the scene: the post-passes wait for the next dtsim_render)
  return field_xfer(h, field, const_cast<void*>(src), bytes, false);
}

void* dtsim_field_devptr(dtsim_t* h, int field) {
  if (!h) return nullptr;
  if (field == DTSIM_FIELD_STATE_BLOB) return h->slab;
  if (field == DTSIM_FIELD_POS) return h->A.pos_x;  // planar: x plane; z plane = pos_z
  FieldDesc d;
  if (!field_desc(h, field, d)) return nullptr;
  return d.base;
}

size_t dtsim_field_bytes(const dtsim_t* h, int field) { return h ? public_bytes(h, field) : 0; }
size_t dtsim_state_bytes(const dtsim_t* h) { return h ? h->slab_bytes : 0; }

int dtsim_sync(dtsim_t* h) {
  if (!h) return fail(DTSIM_E_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DTSIM_OK;
}

void* dtsim_stream(dtsim_t* h) { return h ? (void*)h->stream : nullptr; }

int dtsim_profile_read(dtsim_t* h, int kernel, int* n_launches, double* total_ms) {
  if (!h || kernel < 0 || kernel >= DTSIM_KERNEL__COUNT || !n_launches || !total_ms) return fail(DTSIM_E_INVALID, "bad argument");
  if (!(h->cfg.flags & DTSIM_F_PROFILE)) return fail(DTSIM_E_STATE, "handle created without DTSIM_F_PROFILE");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  ProfSlot& s = h->prof[kernel];
  double tot = 0;
  for (auto& p : s.pending) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
    tot += ms;
    s.free_.push_back(p);
  }
  *n_launches = (int)s.pending.size();
  *total_ms = tot;
  s.pending.clear();
  return DTSIM_OK;
}

}  // extern "C"
