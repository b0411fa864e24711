// C ABI of libsphmi.so (include/sphmi.h): solver lifetime, the 18 stage entry points that mirror
// owOpenCLSolver::_run* (owOpenCLSolver.cpp:213-687), the fused step, read-back and reference-layout export.
// There is no CPU path in this library: every entry point needs a HIP device and fails with SPH_ERR_HIP otherwise.
#include <stdarg.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "sph_common.h"
#include "sph_fastmath.h"

#define SPH_STR_(x) #x
#define SPH_STR(x) SPH_STR_(x)
static thread_local char g_err[512] = "";
void sph_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sph_last_error(void) { return g_err; }
extern "C" int sph_abi_version(void) { return SPHMI_ABI_VERSION; }

// What this binary was built as. Diagnostic / timing-only variants (make variant EXTRA=-D...) compute INVALID results; a
// benchmark must not load one unnoticed (bench.py prints this string and refuses a build whose info contains "DIAG").
extern "C" const char* sph_build_info(void) {
  return "libsphmi gfx950 abi " SPH_STR(SPHMI_ABI_VERSION)
#ifdef DIAG_OWN_GATHER
         " DIAG_OWN_GATHER(invalid results)"
#endif
#ifdef DIAG_NO_WALK
         " DIAG_NO_WALK(invalid results)"
#endif
#ifdef DIAG_DENSITY_INTO_REC
         " DIAG_DENSITY_INTO_REC(invalid results)"
#endif
#ifdef DIAG_NO_REPLAY
         " DIAG_NO_REPLAY(invalid results)"
#endif
#ifdef FN_STAMPS
         " FN_STAMPS"
#endif
#ifdef NO_XCD_REMAP
         " NO_XCD_REMAP"
#endif
#ifdef SPH_VARIANT
         " variant:" SPH_STR(SPH_VARIANT)
#endif
      ;
}

// stage progress bits for the order contract of simulationStep()
enum { P_HASH = 1, P_SORT = 2, P_SORTPOST = 4, P_INDEXX = 8, P_INDEXPOST = 16, P_FIND = 32, P_DENSITY = 64, P_FORCES = 128,
       P_PREDICTPOS = 256, P_PREDICTDENS = 512, P_PRESSUREFORCE = 1024 };

#define NEED(s, bits, what)                                                              \
  do {                                                                                   \
    if (!(s)) { sph_set_error("null solver"); return SPH_ERR_INVALID; }                  \
    if (((s)->progress & (bits)) != (bits)) {                                            \
      sph_set_error("%s called before the stage(s) it depends on (simulationStep order, " \
                    "owPhysicsFluidSimulator.cpp:88-113)", what);                        \
      return SPH_ERR_ORDER;                                                              \
    }                                                                                    \
  } while (0)

template <typename T>
static int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  SPH_HIP(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
  return SPH_OK;
}

SphDev sph_ranged(const sph_solver* s, int ghostDepth) {
  SphDev d = s->d;
  d.rangeLo = 0; d.rangeHi = d.G;
  if (s->hasSlab && ghostDepth >= 0) {
    const long long layerCells = (long long)d.gx * d.gy;
    long long lo = (long long)s->slab.layerLo - ghostDepth, hi = (long long)s->slab.layerHi + ghostDepth;
    if (lo < 0) lo = 0;
    if (hi > d.gz) hi = d.gz;
    if (hi < lo) hi = lo;
    d.rangeLo = (int)(lo * layerCells); d.rangeHi = (int)(hi * layerCells);
  }
  return d;
}

SphDev sph_ranged_layers(const sph_solver* s, long long lo, long long hi) {
  SphDev d = s->d;
  const long long layerCells = (long long)d.gx * d.gy;
  if (lo < 0) lo = 0;
  if (hi > d.gz) hi = d.gz;
  if (hi < lo) hi = lo;
  d.rangeLo = (int)(lo * layerCells); d.rangeHi = (int)(hi * layerCells);
  return d;
}

static int bit_length(uint32_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

// Radial-histogram bin of a squared distance exactly as findNeighbors computes it (sphFluid.cl:159-160):
// (int)(sqrt(d2) * radius_segments / h), float arithmetic, IEEE sqrt and divide (this file is built with -ffp-contract=off).
static int radial_bin(float d2, float h) {
  volatile float dist = sqrtf(d2);
  volatile float scaled = dist * (float)SPH_RSEG;
  volatile float q = scaled / h;
  return (int)q;
}
static float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// binU[j], j = 0..29: the smallest float U such that "d2 < U" <=> "d2 <= h*h and radial_bin(d2) <= j". radial_bin is
// monotone in d2 (sqrt, multiply and divide round monotonically), so T[b] = min{ d2 : radial_bin(d2) >= b } is found by
// bisection over the float bit patterns; U[j] = min(T[j+1], nextafter(h*h)).
static void compute_bin_thresholds(float h, float* U) {
  volatile float h2v = h * h;
  const float h2 = h2v;
  const uint32_t h2next = float_to_bits(h2) + 1u;
  const uint32_t top = float_to_bits(16.0f * h2);
  for (int j = 0; j < SPH_RSEG; j++) {
    uint32_t lo = 0u, hi = top;  // smallest bit pattern with radial_bin >= j+1
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if (radial_bin(bits_to_float(mid), h) >= j + 1) hi = mid; else lo = mid + 1u;
    }
    U[j] = bits_to_float(lo < h2next ? lo : h2next);
  }
  U[SPH_RSEG] = U[SPH_RSEG + 1] = bits_to_float(h2next);
  // U[32 + jb], jb = 0..30: r_thr^2 of pass 1, r_thr = (float)(jb + 1) * h / (float)radius_segments evaluated in float like
  // sphFluid.cl:313-321 (jb = 30: "fewer than 32 candidates", r_thr = 31h/30)
  for (int jb = 0; jb <= SPH_RSEG; jb++) {
    volatile float a = (float)(jb + 1) * h;
    volatile float r = a / (float)SPH_RSEG;
    volatile float r2 = r * r;
    U[32 + jb] = r2;
  }
  // U[63]: radius^2 of the search kernel's filter, a superset of pass 0 (h) and of every pass-1 radius (<= 31h/30)
  {
    const float rmax2 = U[32 + SPH_RSEG] > h2 ? U[32 + SPH_RSEG] : h2;
    volatile float f = rmax2 * (1.f + 0x1p-20f);
    U[63] = f;
  }
}

// ---------------------------------------------------------------------------------------------- timing
struct StageTimer {
  sph_solver* s; int stage; hipEvent_t a, b; bool on;
  StageTimer(sph_solver* s_, int stage_) : s(s_), stage(stage_), a(nullptr), b(nullptr), on(s_->timing) {
    if (!on) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    hipEventRecord(a, s->stream);
  }
  ~StageTimer() {
    if (!on) return;
    hipEventRecord(b, s->stream);
    if (s->numPending == s->capPending) {
      s->capPending = s->capPending ? s->capPending * 2 : 256;
      s->pending = (sph_solver::Pending*)realloc(s->pending, sizeof(sph_solver::Pending) * s->capPending);
    }
    s->pending[s->numPending++] = {stage, a, b};
  }
};

static int resolve_pending(sph_solver* s) {
  if (s->numPending == 0) return SPH_OK;
  SPH_HIP(hipStreamSynchronize(s->stream));
  for (int i = 0; i < s->numPending; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->pending[i].a, s->pending[i].b) == hipSuccess) {
      s->stageMs[s->pending[i].stage] += (double)ms;
      s->stageLaunches[s->pending[i].stage] += 1;
    }
    hipEventDestroy(s->pending[i].a);
    hipEventDestroy(s->pending[i].b);
  }
  s->numPending = 0;
  return SPH_OK;
}

extern "C" int sph_set_stage_timing(sph_solver* s, int enable) {
  if (!s) return SPH_ERR_INVALID;
  s->timing = enable != 0;
  return SPH_OK;
}
extern "C" int sph_reset_stage_times(sph_solver* s) {
  if (!s) return SPH_ERR_INVALID;
  int rc = resolve_pending(s);
  // (every diagnostic counter except dbg[6], the non-finite-coordinate count that check_finite_state reports)
  hipMemsetAsync(s->d.dbg, 0, sizeof(uint32_t) * 6, s->stream);
  hipMemsetAsync(s->d.dbg + 7, 0, sizeof(uint32_t) * (SPH_DBG_WORDS - 7), s->stream);
  memset(s->stageMs, 0, sizeof(s->stageMs));
  memset(s->stageLaunches, 0, sizeof(s->stageLaunches));
  return rc;
}
extern "C" int sph_get_stage_times(sph_solver* s, double* ms_total, int64_t* launches, int n) {
  if (!s || n != SPH_ST_COUNT) return SPH_ERR_INVALID;
  int rc = resolve_pending(s);
  if (rc) return rc;
  for (int i = 0; i < n; i++) {
    if (ms_total) ms_total[i] = s->stageMs[i];
    if (launches) launches[i] = s->stageLaunches[i];
  }
  return SPH_OK;
}

// ---------------------------------------------------------------------------------------------- create / destroy
// Two different libamdhip64 files in one process (e.g. /opt/rocm's and the copy bundled with PyTorch) do not both see the
// GPU. Returns true and the two paths if /proc/self/maps shows that situation.
static bool two_hip_runtimes(char* out, size_t cap) {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return false;
  char line[1024], first[512] = "";
  bool two = false;
  while (!two && fgets(line, sizeof(line), f)) {
    if (!strstr(line, "libamdhip64")) continue;
    char* path = strchr(line, '/');
    if (!path) continue;
    path[strcspn(path, "\n")] = 0;
    if (!first[0]) snprintf(first, sizeof(first), "%s", path);
    else if (strcmp(first, path) != 0) { snprintf(out, cap, "%s and %s", first, path); two = true; }
  }
  fclose(f);
  return two;
}

static void free_all(sph_solver* s) {
  SphDev& d = s->d;
  void* ptrs[] = {d.elasticMask, d.bndMask, d.rp, d.gatherRec, d.posOrig, d.velOrig, d.membDelta, d.sortedPos, d.sortedVel, d.predPos, d.acc, d.accP, d.keys, d.vals,
                  d.keysAlt, d.valsAlt, d.backIndex, d.cellStart, d.cellStartRaw, d.nbrId, d.nbrDist, d.nbr16, d.nbrBase, d.rho,
                  d.elastic, d.membraneData, d.pml, d.muscle, d.dbg, (void*)d.binU, d.gid, d.owned, s->slabCounts,
                  s->blockHist};
  for (void* p : ptrs) if (p) hipFree(p);
  if (s->slabHost) hipHostFree(s->slabHost);
  for (int i = 0; i < s->numHostRegs; i++) hipHostUnregister(s->hostRegs[i].p);
  s->numHostRegs = 0;
  if (s->copyStage) hipHostFree(s->copyStage);
  if (s->pinnedFlags) hipHostFree(s->pinnedFlags);
  if (s->evReadReady) hipEventDestroy(s->evReadReady);
  if (s->evCopyDone) hipEventDestroy(s->evCopyDone);
  if (s->copyStream) hipStreamDestroy(s->copyStream);
  if (s->slabMsgEvent) hipEventDestroy(s->slabMsgEvent);
  if (s->slabRebuildEvent) hipEventDestroy(s->slabRebuildEvent);
  if (s->ownStream && s->stream) hipStreamDestroy(s->stream);
  free(s->pending);
  free(s->hostScratch);
}

extern "C" int sph_destroy(sph_solver* s) {
  if (!s) return SPH_OK;
  hipSetDevice(s->cfg.device);
  if (s->stream) hipStreamSynchronize(s->stream);
  if (s->copyStream) hipStreamSynchronize(s->copyStream);
  for (int i = 0; i < s->numPending; i++) { hipEventDestroy(s->pending[i].a); hipEventDestroy(s->pending[i].b); }
  s->numPending = 0;
  free_all(s);
  delete s;
  return SPH_OK;
}

extern "C" int sph_create(const sph_config* cfg, const float* position, const float* velocity, const float* elastic,
                          const int32_t* membraneData, const int32_t* pml, sph_solver** out) {
  if (!cfg || !position || !velocity || !out) { sph_set_error("sph_create: null argument"); return SPH_ERR_INVALID; }
  *out = nullptr;
  if (cfg->abi_version != SPHMI_ABI_VERSION) { sph_set_error("sph_config.abi_version %d != %d", cfg->abi_version, SPHMI_ABI_VERSION); return SPH_ERR_INVALID; }
  const int N = cfg->particleCount;
  const int cap = cfg->capacity > 0 ? cfg->capacity : N;
  // (k_find_neighbors forms 32-bit element indices into the tiled neighbour map, 32 slots per particle: capacity * 32 < 2^32)
  static_assert((long long)SPH_MAX_PARTICLES * SPH_MAXN < 0x100000000LL, "uint32 mapBase in k_find_neighbors");
  if (N <= 0 || cap < N || cap > SPH_MAX_PARTICLES) {
    sph_set_error("particleCount %d / capacity %d out of range: one solver holds at most %d particles (2^27 - 1; 32-bit neighbour-map "
                  "indices)", N, cap, SPH_MAX_PARTICLES);
    return SPH_ERR_INVALID;
  }
  if (cfg->gridCellsX <= 0 || cfg->gridCellsY <= 0 || cfg->gridCellsZ <= 0 ||
      (long long)cfg->gridCellsX * cfg->gridCellsY * cfg->gridCellsZ != (long long)cfg->gridCellCount) {
    sph_set_error("gridCellCount does not equal gridCellsX*gridCellsY*gridCellsZ");
    return SPH_ERR_INVALID;
  }
  if (cfg->cellIdMask != 0xffffu && cfg->cellIdMask != 0xffffffffu) { sph_set_error("cellIdMask must be 0xffff (reference) or 0xffffffff (wide)"); return SPH_ERR_INVALID; }
  if (cfg->numOfElasticP < 0 || cfg->numOfElasticP + cfg->elasticOffset > N || (cfg->numOfElasticP > 0 && !elastic)) {
    sph_set_error("elastic configuration inconsistent");
    return SPH_ERR_INVALID;
  }
  if (cfg->numOfElasticP > 0 && (cfg->muscleCount <= 0 || cfg->muscleCount > 4096)) { sph_set_error("muscleCount out of range"); return SPH_ERR_INVALID; }
  if (cfg->maxIteration < 1) { sph_set_error("maxIteration must be >= 1"); return SPH_ERR_INVALID; }
  if (!(cfg->h > 0.f) || !(cfg->hashGridCellSize > 0.f)) { sph_set_error("h / hashGridCellSize must be positive"); return SPH_ERR_INVALID; }

  uint32_t liquidSig = 0u;
  // Input sanity the reference does not have. Non-finite coordinates make every particle hash into one cell (a quadratic
  // search); in wide mode a coordinate outside the box gives a cell id outside [0, gridCellCount), and the radix sort only
  // orders the bits a valid id can have. (Reference mode keeps the reference's behaviour for out-of-box input: ids alias.)
  // Creation is the only way in for such coordinates: integrate clamps every new position into the box (sphFluid.cl:1750-1755,
  // integrate_particle), and a NaN — which no clamp catches — is counted by the hash kernel (dbg[6], check_finite_state).
  {
    const bool wide = cfg->cellIdMask == 0xffffffffu;
    for (int i = 0; i < N; i++) {
      if ((int)position[4 * (size_t)i + 3] != SPH_BOUNDARY_PARTICLE && liquidSig != 0xffffffffu) {  // (see sph_slab_liquid_signature)
        uint32_t tb, wb;
        memcpy(&tb, &position[4 * (size_t)i + 3], 4); memcpy(&wb, &velocity[4 * (size_t)i + 3], 4);
        if (wb != 0u || tb == 0u || tb == 0xffffffffu || (liquidSig != 0u && liquidSig != tb)) liquidSig = 0xffffffffu;
        else liquidSig = tb;
      }
      const float x = position[4 * (size_t)i], y = position[4 * (size_t)i + 1], z = position[4 * (size_t)i + 2];
      const bool finite = std::isfinite(x) && std::isfinite(y) && std::isfinite(z);
      const bool inside = x >= cfg->xmin && x <= cfg->xmax && y >= cfg->ymin && y <= cfg->ymax && z >= cfg->zmin && z <= cfg->zmax;
      if (!finite || (wide && !inside)) {
        sph_set_error("particle %d at (%g, %g, %g) is %s", i, x, y, z, finite ? "outside the box (wide cell ids need in-box input)" : "not finite");
        return SPH_ERR_INVALID;
      }
    }
  }
  int ndev = 0;
  const hipError_t devErr = hipGetDeviceCount(&ndev);
  if (devErr != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    char two[300];
    if (two_hip_runtimes(two, sizeof(two)))  // the usual cause of "no ROCm-capable device" on a box that has one
      sph_set_error("no HIP device %d (found %d): two HIP runtimes are mapped into this process (%s); load the one the rest of the "
                    "process uses BEFORE libsphmi.so so that both bind to it", cfg->device, ndev, two);
    else
      sph_set_error("no HIP device %d (found %d%s%s): libsphmi has no CPU fallback", cfg->device, ndev,
                    devErr != hipSuccess ? ", " : "", devErr != hipSuccess ? hipGetErrorString(devErr) : "");
    return SPH_ERR_HIP;
  }
  SPH_HIP(hipSetDevice(cfg->device));

  sph_solver* s = new sph_solver();
  memset((void*)s, 0, sizeof(*s));
  s->cfg = *cfg;
  s->liquidSig = liquidSig;
  if (cfg->stream) { s->stream = (hipStream_t)cfg->stream; s->ownStream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { sph_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete s; return SPH_ERR_HIP; }
    s->ownStream = true;
  }
  SphDev& d = s->d;
  d.N = N; d.G = cfg->gridCellCount; d.gx = cfg->gridCellsX; d.gy = cfg->gridCellsY; d.gz = cfg->gridCellsZ;
  d.cellMask = cfg->cellIdMask;
  d.h = cfg->h; d.cellSize = cfg->hashGridCellSize; d.cellSizeInv = cfg->hashGridCellSizeInv;
  d.simScale = cfg->simulationScale; d.simScaleInv = cfg->simulationScaleInv;
  d.xmin = cfg->xmin; d.xmax = cfg->xmax; d.ymin = cfg->ymin; d.ymax = cfg->ymax; d.zmin = cfg->zmin; d.zmax = cfg->zmax;
  d.r0 = cfg->r0; d.mass = cfg->mass; d.rho0 = cfg->rho0; d.dt = cfg->timeStep; d.delta = cfg->delta;
  d.gravx = cfg->gravity_x; d.gravy = cfg->gravity_y; d.gravz = cfg->gravity_z;
  d.surfTens = cfg->surfTensCoeff;
  // per-step constants in the reference's exact expression types (see sph_common.h for the source lines)
  {
    volatile float massMu = cfg->mass * cfg->viscosity;
    volatile float hs = cfg->h * cfg->simulationScale;
    volatile float hs2 = hs * hs;
    volatile float hs4 = hs2 * hs2;
    volatile float hs6 = hs4 * hs2;
    volatile float pts = cfg->timeStep * cfg->simulationScaleInv;
    volatile float r0d = cfg->rho0 * cfg->delta;
    volatile float halfHs = hs / 2;
    d.massMu = massMu; d.hs = hs; d.hs2 = hs2; d.hs6 = hs6; d.posTimeStep = pts; d.rho0delta = r0d;
    d.closeR = 0.5 * (double)halfHs;
    // (double)r < closeR  <=>  r < closeRf, with closeRf the smallest float whose double value is >= closeR
    float f = (float)d.closeR;
    while ((double)f < d.closeR) f = nextafterf(f, INFINITY);
    while ((double)nextafterf(f, -INFINITY) >= d.closeR) f = nextafterf(f, -INFINITY);
    d.closeRf = f;
  }
  d.massWpoly6 = ((double)cfg->mass) * cfg->Wpoly6Coefficient;
  // operand bounds of k_pressure_force's short division / square-root path; the smallest density a kernel can write is hs^6 * mass * Wpoly6
  sph_fast_bounds(cfg->simulationScale, d.hs, (float)((double)d.hs6 * d.massWpoly6), &d.fastD2Min, &d.fastD2Max, &d.fastValueMin);
  d.massGradW = ((double)cfg->mass) * cfg->gradWspikyCoefficient;
  d.del2W = cfg->del2WviscosityCoefficient;
  d.numElastic = cfg->numOfElasticP; d.elasticOffset = cfg->elasticOffset; d.muscleCount = cfg->muscleCount;
  d.numMembranes = cfg->numOfMembranes; d.hasElastic = cfg->numOfElasticP > 0;
  d.rangeLo = 0; d.rangeHi = d.G;

  s->capacity = cap;
  s->capTiles = (cap + SPH_TILE - 1) / SPH_TILE;
  s->sortBits = (cfg->cellIdMask == 0xffffu) ? 16 : bit_length((uint32_t)(d.G > 0 ? d.G - 1 : 0));
  if (s->sortBits < 1) s->sortBits = 1;
  s->maxSortBlocks = (cap + (SPH_BLOCK * 16) - 1) / (SPH_BLOCK * 16);

  int rc = SPH_OK;
  const size_t n = (size_t)cap, nUp = (size_t)N, G1 = (size_t)d.G + 1, mapN = (size_t)s->capTiles * 64 * 32;
#define A(ptr, count) if (rc == SPH_OK) rc = dev_alloc(&(ptr), (count))
  A(d.posOrig, n); A(d.velOrig, n); A(d.sortedPos, n); A(d.sortedVel, n); A(d.predPos, 3 * n); A(d.acc, n); A(d.accP, n); A(d.rp, n); A(d.bndMask, n); A(d.gatherRec, 2 * ((n + 3) / 4 * 4));
  A(d.keys, n); A(d.vals, n); A(d.keysAlt, n); A(d.valsAlt, n); A(d.backIndex, n);
  A(d.cellStart, G1); A(d.cellStartRaw, G1);
  A(d.nbrId, mapN); A(d.nbrDist, mapN); A(d.nbr16, mapN); A(d.nbrBase, (size_t)s->capTiles * 64);
  A(d.rho, n);
  A(s->blockHist, (size_t)SPH_SORT_MAX_DIGITS * s->maxSortBlocks + SPH_SORT_MAX_DIGITS);  // block histograms + digit totals
  A(d.gid, n); A(d.owned, n); A(s->slabCounts, SPH_SLAB_COUNT_WORDS);
  A(d.dbg, SPH_DBG_WORDS);
  float* binU = nullptr;
  A(binU, 64);
  d.binU = binU;
  if (d.hasElastic) {
    A(d.elasticMask, n); A(d.membDelta, n); A(d.elastic, (size_t)32 * d.numElastic); A(d.muscle, (size_t)d.muscleCount);
    if (membraneData && pml && cfg->numOfMembranes > 0) { A(d.membraneData, (size_t)3 * cfg->numOfMembranes); A(d.pml, (size_t)7 * d.numElastic); }
  }
#undef A
  if (rc != SPH_OK) { free_all(s); delete s; return rc; }

#define UP(dst, src, bytes) do { hipError_t e_ = hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, s->stream); \
    if (e_ != hipSuccess) { sph_set_error("upload failed: %s", hipGetErrorString(e_)); free_all(s); delete s; return SPH_ERR_HIP; } } while (0)
  UP(d.posOrig, position, sizeof(float4) * nUp);
  UP(d.velOrig, velocity, sizeof(float4) * nUp);
  if (d.hasElastic) {
    UP(d.elastic, elastic, sizeof(float4) * 32 * (size_t)d.numElastic);
    // quirk #18: the reference never uploads the signal before step 0 (zeros in practice)
    hipMemsetAsync(d.muscle, 0, sizeof(float) * (size_t)d.muscleCount, s->stream);
    hipMemsetAsync(d.membDelta, 0, sizeof(float4) * n, s->stream);
    if (d.membraneData) {
      UP(d.membraneData, membraneData, sizeof(int32_t) * 3 * (size_t)cfg->numOfMembranes);
      UP(d.pml, pml, sizeof(int32_t) * 7 * (size_t)d.numElastic);
    }
  }
  {
    float U[64];
    compute_bin_thresholds(cfg->h, U);
    UP(binU, U, sizeof(U));
  }
#undef UP
  // buffers the reference leaves uninitialised but that an export may read before they are written
  hipMemsetAsync(d.predPos, 0, sizeof(float) * 3 * n, s->stream);
  hipMemsetAsync(d.acc, 0, sizeof(float4) * n, s->stream);
  hipMemsetAsync(d.accP, 0, sizeof(float4) * n, s->stream);
  hipMemsetAsync(d.rho, 0, sizeof(float) * n, s->stream);
  hipMemsetAsync(d.rp, 0, sizeof(float2) * n, s->stream);
  hipMemsetAsync(d.nbrId, 0xff, sizeof(int32_t) * mapN, s->stream);
  hipMemsetAsync(d.nbrDist, 0, sizeof(float) * mapN, s->stream);
  hipMemsetAsync(d.nbr16, 0xff, sizeof(uint16_t) * mapN, s->stream);
  hipMemsetAsync(d.nbrBase, 0, sizeof(int32_t) * (size_t)s->capTiles * 64, s->stream);
  hipMemsetAsync(d.dbg, 0, sizeof(uint32_t) * SPH_DBG_WORDS, s->stream);
  hipMemsetAsync(d.cellStartRaw, 0, sizeof(uint32_t) * G1, s->stream);
  hipMemsetAsync(d.cellStart, 0, sizeof(uint32_t) * G1, s->stream);
  hipMemsetAsync(d.sortedPos, 0, sizeof(float4) * n, s->stream);
  hipMemsetAsync(d.sortedVel, 0, sizeof(float4) * n, s->stream);
  hipMemsetAsync(d.keys, 0, sizeof(uint32_t) * n, s->stream);
  hipMemsetAsync(d.vals, 0, sizeof(uint32_t) * n, s->stream);
  hipMemsetAsync(d.backIndex, 0, sizeof(uint32_t) * n, s->stream);
  hipError_t e = hipStreamSynchronize(s->stream);  // host arrays may be freed by the caller after return
  if (e != hipSuccess) { sph_set_error("sph_create: %s", hipGetErrorString(e)); free_all(s); delete s; return SPH_ERR_HIP; }
  *out = s;
  return SPH_OK;
}

// ---------------------------------------------------------------------------------------------- stages
static int slab_finish(sph_solver* s, int32_t counts[4]);
// (a rebuild whose particle count is still on its way to the host — sph_slab_rebuild_framed — is finished first)
#define ENTER_RAW(s) do { if (!(s)) { sph_set_error("null solver"); return SPH_ERR_INVALID; } SPH_HIP(hipSetDevice((s)->cfg.device)); } while (0)
#define ENTER(s) do { ENTER_RAW(s); if ((s)->slabRebuildPending) { const int rcf_ = slab_finish((s), nullptr); if (rcf_ != SPH_OK) return rcf_; } } while (0)

extern "C" int sph_run_clear_buffers(sph_solver* s) {
  ENTER(s);
  StageTimer t(s, SPH_ST_FIND_NEIGHBORS);
  return sphk_clear_neighbors(s);
}
extern "C" int sph_run_hash_particles(sph_solver* s) {
  ENTER(s);
  StageTimer t(s, SPH_ST_HASH);
  int rc = sphk_hash(s);
  if (rc == SPH_OK) s->progress = P_HASH;  // a new step starts here
  return rc;
}
extern "C" int sph_run_sort(sph_solver* s) {
  ENTER(s); NEED(s, P_HASH, "sph_run_sort");
  StageTimer t(s, SPH_ST_SORT);
  int rc = sphk_sort(s);
  if (rc == SPH_OK) s->progress |= P_SORT;
  return rc;
}
extern "C" int sph_run_sort_post_pass(sph_solver* s) {
  ENTER(s); NEED(s, P_SORT, "sph_run_sort_post_pass");
  StageTimer t(s, SPH_ST_SORT_POST);
  int rc = sphk_sort_post(s);
  if (rc == SPH_OK) s->progress |= P_SORTPOST;
  return rc;
}
extern "C" int sph_run_indexx(sph_solver* s) {
  ENTER(s); NEED(s, P_SORT, "sph_run_indexx");
  StageTimer t(s, SPH_ST_INDEX);
  int rc = sphk_index_raw(s);
  if (rc == SPH_OK) s->progress |= P_INDEXX;
  return rc;
}
extern "C" int sph_run_index_post_pass(sph_solver* s) {
  ENTER(s); NEED(s, P_INDEXX, "sph_run_index_post_pass");
  StageTimer t(s, SPH_ST_INDEX);
  int rc = sphk_index_fixed(s);
  if (rc == SPH_OK) s->progress |= P_INDEXPOST;
  return rc;
}
extern "C" int sph_run_find_neighbors(sph_solver* s) {
  ENTER(s); NEED(s, P_SORTPOST | P_INDEXPOST, "sph_run_find_neighbors");
  StageTimer t(s, SPH_ST_FIND_NEIGHBORS);
  int rc = sphk_find_neighbors(s);
  if (rc == SPH_OK) s->progress |= P_FIND;
  return rc;
}
extern "C" int sph_run_pcisph_compute_density(sph_solver* s) {
  ENTER(s); NEED(s, P_FIND, "sph_run_pcisph_compute_density");
  StageTimer t(s, SPH_ST_DENSITY);
  int rc = sphk_density(s);
  if (rc == SPH_OK) s->progress |= P_DENSITY;
  return rc;
}
extern "C" int sph_run_pcisph_compute_forces_and_init_pressure(sph_solver* s) {
  ENTER(s); NEED(s, P_DENSITY, "sph_run_pcisph_compute_forces_and_init_pressure");
  StageTimer t(s, SPH_ST_FORCES);
  int rc = sphk_forces(s, false);
  if (rc == SPH_OK) s->progress |= P_FORCES;
  return rc;
}
extern "C" int sph_run_pcisph_compute_elastic_forces(sph_solver* s) {
  ENTER(s); NEED(s, P_FORCES, "sph_run_pcisph_compute_elastic_forces");
  StageTimer t(s, SPH_ST_ELASTIC);
  return sphk_elastic(s);
}
extern "C" int sph_run_pcisph_predict_positions(sph_solver* s) {
  ENTER(s); NEED(s, P_FORCES, "sph_run_pcisph_predict_positions");
  StageTimer t(s, SPH_ST_PRESSURE_FORCE);
  int rc = sphk_predict_positions(s);
  if (rc == SPH_OK) s->progress |= P_PREDICTPOS;
  return rc;
}
extern "C" int sph_run_pcisph_predict_density(sph_solver* s) {
  ENTER(s); NEED(s, P_PREDICTPOS, "sph_run_pcisph_predict_density");
  StageTimer t(s, SPH_ST_PREDICT_DENSITY);
  int rc = sphk_predict_density(s, false);
  if (rc == SPH_OK) s->progress |= P_PREDICTDENS;
  return rc;
}
extern "C" int sph_run_pcisph_correct_pressure(sph_solver* s) {
  ENTER(s); NEED(s, P_PREDICTDENS, "sph_run_pcisph_correct_pressure");
  StageTimer t(s, SPH_ST_PREDICT_DENSITY);
  return sphk_correct_pressure(s);
}
extern "C" int sph_run_pcisph_compute_pressure_force_acceleration(sph_solver* s) {
  ENTER(s); NEED(s, P_PREDICTDENS, "sph_run_pcisph_compute_pressure_force_acceleration");
  StageTimer t(s, SPH_ST_PRESSURE_FORCE);
  int rc = sphk_pressure_force(s, 0);
  if (rc == SPH_OK) s->progress |= P_PRESSUREFORCE;
  return rc;
}
extern "C" int sph_run_pcisph_integrate(sph_solver* s, int iterationCount) {
  (void)iterationCount;  // only used by commented-out debug prints in the reference (sphFluid.cl:1784-1805)
  ENTER(s); NEED(s, P_FORCES, "sph_run_pcisph_integrate");
  StageTimer t(s, SPH_ST_INTEGRATE);
  return sphk_integrate(s);
}
extern "C" int sph_run_clear_membrane_buffers(sph_solver* s) {
  ENTER(s);
  StageTimer t(s, SPH_ST_MEMBRANES);
  return sphk_clear_membranes(s);
}
extern "C" int sph_run_compute_interaction_with_membranes(sph_solver* s) {
  ENTER(s); NEED(s, P_FIND, "sph_run_compute_interaction_with_membranes");
  if (!s->d.pml) return SPH_OK;  // no membrane lists were supplied
  StageTimer t(s, SPH_ST_MEMBRANES);
  return sphk_membranes(s);
}
extern "C" int sph_run_compute_interaction_with_membranes_finalize(sph_solver* s) {
  ENTER(s);
  StageTimer t(s, SPH_ST_MEMBRANES);
  return sphk_membranes_finalize(s);
}

// The fused fast path. Stage sequence of simulationStep() (owPhysicsFluidSimulator.cpp:88-113) with:
//   clearBuffers folded into findNeighbors; sortPostPass + indexx + index fix-up in one kernel;
//   predictPositions folded into the kernel that produces the pressure acceleration it integrates
//   (forces for iteration 0, pressure force for the later ones); correctPressure folded into predictDensity;
//   integrate folded into the last pressure-force kernel; membrane kernels skipped when there is no elastic matter
//   (their only effect, `position += 0`, is applied in integrate).
// The launches of one fused step, in order, on s->stream. `tail` (slab mode, overlapped step only): the last stage —
// pressure force + integrate, the only one whose results the halo messages carry — is launched first on the owned layers
// next to the cuts, then the messages are packed (tail->packMessages), then the remaining owned layers follow.
struct StepTail {
  uint32_t *frameDown, *frameUp;
  int capRecords;
};

static int enqueue_step(sph_solver* s, const StepTail* tail) {
  int rc;
#define RUN(stage, call) do { StageTimer t_(s, stage); rc = (call); if (rc != SPH_OK) return rc; } while (0)
  if (s->hasSlab) {
    RUN(SPH_ST_SORT, sphk_hash_sort_post_slab(s));
    RUN(SPH_ST_SORT_POST, sphk_index_fixed(s));
  } else {
    int bits = s->sortBits;
    bool compact = false;  // wide cell ids: only ~1/8 of the declared index space is reachable, which can save a radix pass
    RUN(SPH_ST_HASH, sphk_hash_for_step(s, &bits, &compact));
    RUN(SPH_ST_SORT, sphk_sort_pairs(s, s->d.N, bits));
    if (compact) {
      StageTimer t_(s, SPH_ST_SORT_POST);
      rc = sphk_sort_post_rekey(s);
      if (rc == SPH_OK) rc = sphk_index_fixed(s);
      if (rc != SPH_OK) return rc;
    } else {
      RUN(SPH_ST_SORT_POST, sphk_sort_post_and_index(s));
    }
  }
  // Slab mode: a stage runs only on the ghost layers its results are needed on (owned layers + depth layers per side).
  // Information travels one neighbour hop (<= 31h/30, i.e. 31/60 of a 2h cell layer) per stage, backwards from the owned
  // layers: with `left` predict-correct iterations still to come, the pressure force is needed 2*left hops out and
  // predictDensity 2*left + 1; density 1 hop; the neighbour lists as far as the first predictDensity; forces on the owned
  // layers only; the iteration-0 predicted positions one hop further than anything else (k_ghost_init: everywhere).
  const bool slab = s->hasSlab;
  const int M = s->cfg.maxIteration;
  auto layersFor = [&](int hops) { return slab ? min((hops * 31 + 59) / 60, s->slab.ghostLayers) : -1; };
  RUN(SPH_ST_FIND_NEIGHBORS, sphk_find_neighbors(s, layersFor(2 * (M - 1) + 1)));
  RUN(SPH_ST_DENSITY, sphk_density(s, layersFor(1)));
  if (slab) RUN(SPH_ST_FORCES, sphk_ghost_init(s));
  RUN(SPH_ST_FORCES, sphk_forces(s, true, layersFor(0)));
  if (s->d.hasElastic) RUN(SPH_ST_ELASTIC, sphk_elastic(s));
  for (int iter = 0; iter < M; iter++) {
    const int left = M - 1 - iter;  // iterations after this one
    RUN(SPH_ST_PREDICT_DENSITY, sphk_predict_density(s, true, layersFor(2 * left + 1), iter == 0));
    if (left > 0 || !tail) { RUN(SPH_ST_PRESSURE_FORCE, sphk_pressure_force(s, left == 0 ? 2 : 1, layersFor(2 * left))); continue; }
    // ---- overlapped tail. A particle that ends the step within W layers of a cut started it within W + 1 layers (particles
    // move less than one layer per step — the same assumption the ghost depth rests on), so integrating the W + 1 owned
    // layers next to each cut first makes every message particle final; the pack below only looks at those.
    const long long lo = max((long long)s->slab.layerLo, 0LL), hi = min((long long)s->slab.layerHi, (long long)s->d.gz);
    const long long W1 = (long long)s->slab.ghostLayers + 1;
    const long long dEnd = s->slab.hasLower ? min(lo + W1, hi) : lo;         // [lo, dEnd): next to the lower cut
    const long long uBeg = s->slab.hasUpper ? max(hi - W1, dEnd) : hi;       // [uBeg, hi): next to the upper cut
    if (dEnd > lo) RUN(SPH_ST_PRESSURE_FORCE, sphk_pressure_force_layers(s, 2, lo, dEnd));
    if (hi > uBeg) RUN(SPH_ST_PRESSURE_FORCE, sphk_pressure_force_layers(s, 2, uBeg, hi));
    const SphDev dA = sph_ranged_layers(s, lo, dEnd), dB = sph_ranged_layers(s, uBeg, hi);
    // (cell ranges: the pack kernel turns them into sorted-index ranges with the cell table, which lives on the device)
    SlabPart part{SLAB_PART_MESSAGES, dA.rangeLo, dA.rangeHi, dB.rangeLo, dB.rangeHi};
    rc = sphk_slab_pack(s, tail->frameDown ? tail->frameDown + 1 : nullptr, tail->frameUp ? tail->frameUp + 1 : nullptr, tail->capRecords,
                        tail->frameDown, tail->frameUp, part, s->slabCounts + 4);
    if (rc != SPH_OK) return rc;
    SPH_HIP(hipMemcpyAsync(s->slabHost + 4, s->slabCounts + 4, sizeof(uint32_t) * 3, hipMemcpyDeviceToHost, s->stream));
    SPH_HIP(hipEventRecord(s->slabMsgEvent, s->stream));
    if (uBeg > dEnd) RUN(SPH_ST_PRESSURE_FORCE, sphk_pressure_force_layers(s, 2, dEnd, uBeg));
    rc = sphk_slab_pack(s, nullptr, nullptr, 0, nullptr, nullptr, SlabPart{SLAB_PART_KEPT, 0, 0, 0, 0}, s->slabCounts + 8);
    if (rc != SPH_OK) return rc;
    SPH_HIP(hipMemcpyAsync(s->slabHost + 8, s->slabCounts + 8, sizeof(uint32_t) * 3, hipMemcpyDeviceToHost, s->stream));
    SPH_HIP(hipMemcpyAsync(s->slabHost + 3, s->slabCounts + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    SPH_HIP(hipMemcpyAsync(s->slabHost + 7, s->slabCounts + 7, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
  }
  if (s->d.hasElastic) {
    RUN(SPH_ST_MEMBRANES, sphk_clear_membranes(s));
    if (s->d.pml) RUN(SPH_ST_MEMBRANES, sphk_membranes(s));
    RUN(SPH_ST_MEMBRANES, sphk_membranes_finalize(s));
  }
#undef RUN
  s->progress = P_HASH | P_SORT | P_SORTPOST | P_INDEXPOST | P_FIND | P_DENSITY | P_FORCES | P_PREDICTPOS | P_PREDICTDENS |
                P_PRESSUREFORCE;
  return SPH_OK;
}

extern "C" int sph_step(sph_solver* s, int iterationCount) {
  (void)iterationCount;
  ENTER(s);
  return enqueue_step(s, nullptr);
}

extern "C" int sph_update_muscles(sph_solver* s, const float* signal, int n) {
  ENTER(s);
  if (!signal || n != s->cfg.muscleCount) { sph_set_error("sph_update_muscles: n must equal muscleCount"); return SPH_ERR_SIZE; }
  if (!s->d.muscle) return SPH_OK;  // no elastic matter: nothing reads the signal
  SPH_HIP(hipMemcpyAsync(s->d.muscle, signal, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s->stream));
  SPH_HIP(hipStreamSynchronize(s->stream));  // blocking, like the reference's enqueueWriteBuffer(CL_TRUE)
  return SPH_OK;
}

static int check_finite_state(sph_solver* s);

extern "C" int sph_step_sort_passes(sph_solver* s) {
  ENTER(s);
  bool compact;
  return sph_sort_passes(sphk_step_sort_bits(s, &compact));
}

extern "C" int sph_synchronize(sph_solver* s) {
  ENTER(s);
  return check_finite_state(s);  // (synchronises the stream)
}

// ---------------------------------------------------------------------------------------------- read-back
static int d2h(sph_solver* s, void* dst, const void* src, size_t bytes) {
  SPH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
  SPH_HIP(hipStreamSynchronize(s->stream));
  return SPH_OK;
}

// dbg[6]: particles with a non-finite coordinate seen by the hash kernel (the state has blown up). Sticky: once seen, every
// blocking call keeps reporting it until the solver is destroyed — a caller that ignores one SPH_ERR_INVALID does not continue
// silently on NaN state.
static int report_blown_up(sph_solver* s) {
  sph_set_error("%llu particle coordinate(s) were not finite: the simulation state has blown up", (unsigned long long)s->blownUp);
  return SPH_ERR_INVALID;
}
static int check_finite_state(sph_solver* s) {
  uint32_t bad = 0;
  SPH_HIP(hipMemcpyAsync(&bad, s->d.dbg + 6, sizeof(bad), hipMemcpyDeviceToHost, s->stream));
  SPH_HIP(hipStreamSynchronize(s->stream));
  if (bad) {
    s->blownUp += bad;
    SPH_HIP(hipMemsetAsync(s->d.dbg + 6, 0, sizeof(uint32_t), s->stream));
  }
  return s->blownUp ? report_blown_up(s) : SPH_OK;
}

extern "C" int sph_read_position(sph_solver* s, float* out) {
  ENTER(s); if (!out) return SPH_ERR_INVALID;
  const int rc = d2h(s, out, s->d.posOrig, sizeof(float4) * (size_t)s->d.N);
  return rc != SPH_OK ? rc : check_finite_state(s);
}
// ---- asynchronous read_position_buffer. The reference's step always ends with a blocking 16N-byte read
// (owPhysicsFluidSimulator.cpp:115; 264 MB at 16.5 M particles: +45 % on the step when it is waited for). posOrig is written by
// exactly one kernel per step, the last one (integrate; + the membrane finalize pass), so the copy of step t can run on its own
// stream under the search and PCISPH stages of step t+1: copyStream waits for an event recorded on s->stream when the read is
// requested, and the next kernel that writes posOrig waits for the copy's event (sph_guard_position_write).
static int copy_setup(sph_solver* s) {
  if (s->copyStream) return SPH_OK;
  SPH_HIP(hipStreamCreateWithFlags(&s->copyStream, hipStreamNonBlocking));
  SPH_HIP(hipEventCreateWithFlags(&s->evReadReady, hipEventDisableTiming));
  SPH_HIP(hipEventCreateWithFlags(&s->evCopyDone, hipEventDisableTiming));
  SPH_HIP(hipHostMalloc((void**)&s->pinnedFlags, sizeof(uint32_t) * 4, hipHostMallocDefault));
  s->pinnedFlags[0] = 0u;
  return SPH_OK;
}

// true if the DMA engine can write [p, p + bytes) directly: pinned already, or page-locked in place now
static bool host_pinned(sph_solver* s, void* p, size_t bytes) {
  for (int i = 0; i < s->numHostRegs; i++)
    if ((char*)p >= (char*)s->hostRegs[i].p && (char*)p + bytes <= (char*)s->hostRegs[i].p + s->hostRegs[i].bytes) return true;
  unsigned int flags = 0;
  if (hipHostGetFlags(&flags, p) == hipSuccess) return true;  // hipHostMalloc'ed or registered by the caller
  (void)hipGetLastError();
  if (s->numHostRegs >= 8) return false;
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  s->hostRegs[s->numHostRegs].p = p; s->hostRegs[s->numHostRegs].bytes = bytes; s->numHostRegs++;
  return true;
}

extern "C" int sph_read_position_wait(sph_solver* s) {
  ENTER(s);
  if (!s->copyPending) return s->blownUp ? report_blown_up(s) : SPH_OK;
  SPH_HIP(hipEventSynchronize(s->evCopyDone));
  s->copyPending = false;
  if (s->copyViaStage) memcpy(s->copyUserDst, s->copyStage, sizeof(float4) * (size_t)s->d.N);
  if (s->pinnedFlags[0]) {  // (the device counter keeps counting; it is cleared by the next blocking check)
    if (!s->blownUp) s->blownUp = s->pinnedFlags[0];
    return report_blown_up(s);
  }
  return s->blownUp ? report_blown_up(s) : SPH_OK;
}

extern "C" int sph_read_position_async(sph_solver* s, float* out) {
  ENTER(s); if (!out) return SPH_ERR_INVALID;
  int rc = copy_setup(s);
  if (rc != SPH_OK) return rc;
  if (s->copyPending) {  // the previous read must have landed before its staging area / flags are reused (long done in a step loop)
    rc = sph_read_position_wait(s);
    if (rc != SPH_OK) return rc;
  }
  const size_t bytes = sizeof(float4) * (size_t)s->d.N;
  void* dst = out;
  s->copyViaStage = false;
  if (!host_pinned(s, out, bytes)) {
    if (s->copyStageBytes < bytes) {
      if (s->copyStage) hipHostFree(s->copyStage);
      s->copyStage = nullptr; s->copyStageBytes = 0;
      SPH_HIP(hipHostMalloc(&s->copyStage, bytes, hipHostMallocDefault));
      s->copyStageBytes = bytes;
    }
    dst = s->copyStage;
    s->copyViaStage = true;
  }
  s->copyUserDst = out;
  SPH_HIP(hipEventRecord(s->evReadReady, s->stream));
  SPH_HIP(hipStreamWaitEvent(s->copyStream, s->evReadReady, 0));
  SPH_HIP(hipMemcpyAsync(dst, s->d.posOrig, bytes, hipMemcpyDeviceToHost, s->copyStream));
  SPH_HIP(hipMemcpyAsync(s->pinnedFlags, s->d.dbg + 6, sizeof(uint32_t), hipMemcpyDeviceToHost, s->copyStream));
  SPH_HIP(hipEventRecord(s->evCopyDone, s->copyStream));
  s->copyPending = true;
  return SPH_OK;
}

extern "C" int sph_host_unregister(sph_solver* s, void* p) {
  ENTER(s);
  if (s->copyPending) { const int rc = sph_read_position_wait(s); if (rc != SPH_OK && rc != SPH_ERR_INVALID) return rc; }
  for (int i = 0; i < s->numHostRegs; i++)
    if (s->hostRegs[i].p == p) {
      hipHostUnregister(p);
      s->hostRegs[i] = s->hostRegs[--s->numHostRegs];
      return SPH_OK;
    }
  return SPH_OK;  // not one of ours: nothing to do
}

int sph_guard_position_write(sph_solver* s) {
  if (s->copyPending) SPH_HIP(hipStreamWaitEvent(s->stream, s->evCopyDone, 0));
  return SPH_OK;
}

extern "C" int sph_read_velocity(sph_solver* s, float* out) {
  ENTER(s); if (!out) return SPH_ERR_INVALID;
  return d2h(s, out, s->d.velOrig, sizeof(float4) * (size_t)s->d.N);
}
extern "C" int sph_read_density(sph_solver* s, float* out) {
  ENTER(s); if (!out) return SPH_ERR_INVALID;
  return d2h(s, out, s->d.rho, sizeof(float) * (size_t)s->d.N);
}
extern "C" int sph_read_particle_index(sph_solver* s, uint32_t* out) {
  ENTER(s); if (!out) return SPH_ERR_INVALID;
  const size_t n = (size_t)s->d.N;
  std::vector<uint32_t> k(n), v(n);
  int rc = d2h(s, k.data(), s->d.keys, sizeof(uint32_t) * n);
  if (rc == SPH_OK) rc = d2h(s, v.data(), s->d.vals, sizeof(uint32_t) * n);
  if (rc != SPH_OK) return rc;
  for (size_t i = 0; i < n; i++) { out[2 * i] = k[i]; out[2 * i + 1] = v[i]; }
  return SPH_OK;
}

// Export in the reference's layouts (SURVEY table 2.2). Test/inspection path: converts on the host.
extern "C" int sph_read_buffer(sph_solver* s, const char* name, void* out, size_t bytes, size_t* needed) {
  ENTER(s);
  if (!name) return SPH_ERR_INVALID;
  const SphDev& d = s->d;
  const size_t n = (size_t)d.N, G1 = (size_t)d.G + 1;
  const int numTiles = (d.N + SPH_TILE - 1) / SPH_TILE;
  size_t need = 0;
  enum { B_POS, B_VEL, B_SPOS, B_SVEL, B_ACC, B_NMAP, B_NIDS, B_PI, B_PIB, B_GCI, B_GCIF, B_P, B_RHO, B_DBG, B_TRACE } which;
  if (!strcmp(name, "position")) { which = B_POS; need = sizeof(float4) * 2 * n; }
  else if (!strcmp(name, "velocity")) { which = B_VEL; need = sizeof(float4) * 2 * n; }
  else if (!strcmp(name, "sortedPosition")) { which = B_SPOS; need = sizeof(float4) * 2 * n; }
  else if (!strcmp(name, "sortedVelocity")) { which = B_SVEL; need = sizeof(float4) * n; }
  else if (!strcmp(name, "acceleration")) { which = B_ACC; need = sizeof(float4) * 2 * n; }
  else if (!strcmp(name, "neighborMap")) { which = B_NMAP; need = sizeof(float) * 2 * 32 * n; }
  else if (!strcmp(name, "neighborIds")) { which = B_NIDS; need = sizeof(int32_t) * 32 * n; }
  else if (!strcmp(name, "particleIndex")) { which = B_PI; need = sizeof(uint32_t) * 2 * n; }
  else if (!strcmp(name, "particleIndexBack")) { which = B_PIB; need = sizeof(uint32_t) * n; }
  else if (!strcmp(name, "gridCellIndex")) { which = B_GCI; need = sizeof(uint32_t) * G1; }
  else if (!strcmp(name, "gridCellIndexFixedUp")) { which = B_GCIF; need = sizeof(uint32_t) * G1; }
  else if (!strcmp(name, "pressure")) { which = B_P; need = sizeof(float) * n; }
  else if (!strcmp(name, "rho")) { which = B_RHO; need = sizeof(float) * 2 * n; }
  else if (!strcmp(name, "diagnosticTrace")) { which = B_TRACE; need = sizeof(uint32_t) * n; }  // scratch words of diagnostic builds
  else if (!strcmp(name, "debugCounters")) { which = B_DBG; need = sizeof(uint32_t) * SPH_DBG_WORDS; }
  else { sph_set_error("unknown buffer '%s'", name); return SPH_ERR_UNKNOWN_BUFFER; }
  if (needed) *needed = need;
  if (!out) return SPH_OK;
  if (bytes != need) { sph_set_error("buffer '%s' is %zu bytes, caller gave %zu", name, need, bytes); return SPH_ERR_SIZE; }
  int rc = SPH_OK;
  char* o = (char*)out;
  switch (which) {
    case B_POS:
      rc = d2h(s, o, d.posOrig, sizeof(float4) * n);
      if (rc == SPH_OK) { if (d.membDelta) rc = d2h(s, o + sizeof(float4) * n, d.membDelta, sizeof(float4) * n); else memset(o + sizeof(float4) * n, 0, sizeof(float4) * n); }
      break;
    case B_VEL:
      rc = d2h(s, o, d.velOrig, sizeof(float4) * n);
      memset(o + sizeof(float4) * n, 0, sizeof(float4) * n);  // the scratch half is only ever zeroed (App. B #16)
      break;
    case B_SPOS: {
      std::vector<uint32_t> k(n);
      std::vector<float4> sv(n);
      rc = d2h(s, o, d.sortedPos, sizeof(float4) * n);
      if (rc == SPH_OK) {  // the predicted half: packed (x, y, z) on the device; .w is dead data in the reference
        std::vector<float> p3(3 * n);
        rc = d2h(s, p3.data(), d.predPos, sizeof(float) * 3 * n);
        float4* half = (float4*)(o + sizeof(float4) * n);
        if (rc == SPH_OK) for (size_t i = 0; i < n; i++) half[i] = make_float4(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2], 0.f);
      }
      if (rc == SPH_OK) rc = d2h(s, k.data(), d.keys, sizeof(uint32_t) * n);
      if (rc == SPH_OK) rc = d2h(s, sv.data(), d.sortedVel, sizeof(float4) * n);
      if (rc != SPH_OK) break;
      float4* a = (float4*)o;
      for (size_t i = 0; i < n; i++) {
        const int type = (int)a[i].w;
        const float cellf = (float)(int)k[i];  // POSITION_CELL_ID = (float)cellId (sphFluid.cl:461)
        a[i].w = cellf;
        // .w of the predicted half is dead data in the reference: cell id for boundary particles, cell id + posTimeStep *
        // (v.w + dt*a_p.w) otherwise, with a_p.w == 0
        a[n + i].w = (type == SPH_BOUNDARY_PARTICLE) ? cellf : cellf + d.posTimeStep * (sv[i].w + d.dt * 0.f);
      }
    } break;
    case B_SVEL: rc = d2h(s, o, d.sortedVel, sizeof(float4) * n); break;
    case B_ACC:
      rc = d2h(s, o, d.acc, sizeof(float4) * n);
      if (rc == SPH_OK) rc = d2h(s, o + sizeof(float4) * n, d.accP, sizeof(float4) * n);
      break;
    case B_NMAP:
    case B_NIDS: {
      const size_t mapN = (size_t)numTiles * 64 * 32;
      std::vector<int32_t> ids(mapN);
      std::vector<float> dist(mapN);
      std::vector<uint16_t> n16(mapN);
      std::vector<int32_t> nbase((size_t)numTiles * 64);
      rc = d2h(s, ids.data(), d.nbrId, sizeof(int32_t) * mapN);
      if (rc == SPH_OK) rc = d2h(s, n16.data(), d.nbr16, sizeof(uint16_t) * mapN);
      if (rc == SPH_OK) rc = d2h(s, nbase.data(), d.nbrBase, sizeof(int32_t) * nbase.size());
      if (rc == SPH_OK) rc = d2h(s, dist.data(), d.nbrDist, sizeof(float) * mapN);
      if (rc == SPH_OK)  // the ids live in the 16-bit map (sph_common.h); the 32-bit rows only where that could not be written
        for (size_t id = 0; id < n; id++)
          for (int k = 0; k < 32; k++) ids[nbr_index((int)id, k)] = nbr_decode(n16.data(), nbase.data(), ids.data(), (int)id, k);
      if (rc != SPH_OK) break;
      for (size_t id = 0; id < n; id++)
        for (int k = 0; k < 32; k++) {
          const size_t src = nbr_index((int)id, k);
          if (which == B_NIDS) ((int32_t*)o)[id * 32 + k] = ids[src];
          else { ((float*)o)[(id * 32 + k) * 2] = (float)ids[src]; ((float*)o)[(id * 32 + k) * 2 + 1] = dist[src]; }
        }
    } break;
    case B_PI: rc = sph_read_particle_index(s, (uint32_t*)o); break;
    case B_PIB: rc = d2h(s, o, d.backIndex, sizeof(uint32_t) * n); break;
    case B_GCI: rc = d2h(s, o, d.cellStartRaw, sizeof(uint32_t) * G1); break;
    case B_GCIF: rc = d2h(s, o, d.cellStart, sizeof(uint32_t) * G1); break;
    case B_P: {
      std::vector<float2> rp(n);
      rc = d2h(s, rp.data(), d.rp, sizeof(float2) * n);
      if (rc == SPH_OK) for (size_t i = 0; i < n; i++) ((float*)o)[i] = rp[i].y;
    } break;
    case B_TRACE: rc = d2h(s, o, d.valsAlt, sizeof(uint32_t) * n); break;
    case B_DBG: rc = d2h(s, o, d.dbg, sizeof(uint32_t) * SPH_DBG_WORDS); break;
    case B_RHO:
      rc = d2h(s, o, d.rho, sizeof(float) * n);
      if (rc == SPH_OK) {
        std::vector<float2> rp(n);
        rc = d2h(s, rp.data(), d.rp, sizeof(float2) * n);
        if (rc == SPH_OK) for (size_t i = 0; i < n; i++) ((float*)o)[n + i] = rp[i].x;
      }
      break;
  }
  return rc;
}

extern "C" int sph_read_neighbor_rows(sph_solver* s, int32_t first, int32_t count, int32_t* ids, float* dist) {
  ENTER(s);
  if (first < 0 || count < 0 || (long long)first + count > s->d.N) { sph_set_error("sph_read_neighbor_rows: range outside [0, N)"); return SPH_ERR_INVALID; }
  if (count == 0) return SPH_OK;
  const size_t t0 = (size_t)first / SPH_TILE, t1 = ((size_t)first + count + SPH_TILE - 1) / SPH_TILE;  // tiles [t0, t1)
  const size_t words = (t1 - t0) * 64 * 32, base = t0 * 64 * 32;
  std::vector<int32_t> ti;
  std::vector<float> td;
  int rc = SPH_OK;
  std::vector<uint16_t> t16;
  std::vector<int32_t> tb;
  if (ids) {
    ti.resize(words); t16.resize(words); tb.resize((t1 - t0) * 64);
    rc = d2h(s, ti.data(), s->d.nbrId + base, sizeof(int32_t) * words);
    if (rc == SPH_OK) rc = d2h(s, t16.data(), s->d.nbr16 + base, sizeof(uint16_t) * words);
    if (rc == SPH_OK) rc = d2h(s, tb.data(), s->d.nbrBase + t0 * 64, sizeof(int32_t) * tb.size());
  }
  if (rc == SPH_OK && dist) { td.resize(words); rc = d2h(s, td.data(), s->d.nbrDist + base, sizeof(float) * words); }
  if (rc != SPH_OK) return rc;
  const int shift = (int)(t0 * 64);  // the copies start at tile t0: decode with tile-relative particle numbers
  for (int32_t i = 0; i < count; i++)
    for (int k = 0; k < 32; k++) {
      const size_t src = nbr_index(first + i, k) - base;
      if (ids) {
        int j = nbr_decode(t16.data(), tb.data(), ti.data(), first + i - shift, k);
        // (offsets are relative to the particle's own sorted index: undo the tile-relative numbering for the unflagged ones)
        const uint32_t e = t16[src];
        if (t16[nbr_index(first + i - shift, 0)] != SPH_N16_WIDE && e != SPH_N16_EMPTY && !(e & 0x8000u)) j += shift;
        ids[(size_t)i * 32 + k] = j;
      }
      if (dist) dist[(size_t)i * 32 + k] = td[src];
    }
  return SPH_OK;
}

// ---------------------------------------------------------------------------------------------- slab decomposition
extern "C" int sph_particle_count(sph_solver* s) {
  ENTER(s);
  return s->d.N;
}

extern "C" int sph_slab_init(sph_solver* s, const sph_slab* slab, const uint32_t* globalIds) {
  ENTER(s);
  if (!slab || !globalIds) { sph_set_error("sph_slab_init: null argument"); return SPH_ERR_INVALID; }
  if (s->cfg.cellIdMask != 0xffffffffu) { sph_set_error("slab decomposition needs wide cell ids (cellIdMask = 0xffffffff)"); return SPH_ERR_INVALID; }
  if (s->d.hasElastic) { sph_set_error("slab decomposition supports pure-liquid scenes only"); return SPH_ERR_INVALID; }
  if (slab->layerLo >= slab->layerHi || slab->ghostLayers < 1 || slab->globalIdBits < 1 || slab->globalIdBits > 32) { sph_set_error("bad sph_slab"); return SPH_ERR_INVALID; }
  if ((2 * s->cfg.maxIteration * 31 + 59) / 60 > slab->ghostLayers) {  // 2*maxIteration hops of 31h/30 must fit the ghost zone
    sph_set_error("ghostLayers = %d is too thin for maxIteration = %d", slab->ghostLayers, s->cfg.maxIteration);
    return SPH_ERR_INVALID;
  }
  s->slab = *slab; s->hasSlab = true; s->slabKept = -1; s->slabStepPending = false;
  if (!s->slabHost) SPH_HIP(hipHostMalloc((void**)&s->slabHost, sizeof(uint32_t) * SPH_SLAB_COUNT_WORDS, hipHostMallocDefault));
  if (!s->slabMsgEvent) SPH_HIP(hipEventCreateWithFlags(&s->slabMsgEvent, hipEventDisableTiming));
  if (!s->slabRebuildEvent) SPH_HIP(hipEventCreateWithFlags(&s->slabRebuildEvent, hipEventDisableTiming));
  SPH_HIP(hipMemcpyAsync(s->d.gid, globalIds, sizeof(uint32_t) * (size_t)s->d.N, hipMemcpyHostToDevice, s->stream));
  // ownership flags from the initial positions: reuse the rebuild path with nothing received
  SPH_HIP(hipMemcpyAsync(s->d.sortedPos, s->d.posOrig, sizeof(float4) * (size_t)s->d.N, hipMemcpyDeviceToDevice, s->stream));
  SPH_HIP(hipMemcpyAsync(s->d.sortedVel, s->d.velOrig, sizeof(float4) * (size_t)s->d.N, hipMemcpyDeviceToDevice, s->stream));
  SPH_HIP(hipMemcpyAsync(s->d.keys, s->d.gid, sizeof(uint32_t) * (size_t)s->d.N, hipMemcpyDeviceToDevice, s->stream));
  SPH_HIP(hipMemsetAsync(s->slabCounts, 0, sizeof(uint32_t) * SPH_SLAB_COUNT_WORDS, s->stream));
  int rc = sphk_slab_sort_rebuild(s, s->d.N);
  if (rc != SPH_OK) return rc;
  SPH_HIP(hipStreamSynchronize(s->stream));
  return SPH_OK;
}

// An owned particle moved more than one cell layer in a step: the assumption behind the halo depth (include/sphmi.h) is broken
static int slab_motion_error(sph_solver* s, uint32_t n) {
  hipMemsetAsync(s->slabCounts + 7, 0, sizeof(uint32_t), s->stream);
  sph_set_error("%u owned particle(s) moved more than one cell layer in one step: the %d-layer halo no longer guarantees "
                "single-domain results (time step too large for these velocities?)", n, s->slab.ghostLayers);
  return SPH_ERR_INVALID;
}

static int slab_pack(sph_solver* s, uint32_t* msgDown, uint32_t* msgUp, int32_t capRecords, int32_t counts[3], uint32_t* headDown,
                     uint32_t* headUp) {
  int rc = sphk_slab_pack(s, msgDown, msgUp, capRecords, headDown, headUp);
  if (rc != SPH_OK) return rc;
  uint32_t h[8];
  SPH_HIP(hipMemcpyAsync(h, s->slabCounts, sizeof(h), hipMemcpyDeviceToHost, s->stream));
  SPH_HIP(hipStreamSynchronize(s->stream));
  counts[0] = (int32_t)h[0]; counts[1] = (int32_t)h[1]; counts[2] = (int32_t)h[2];
  if (h[3]) {  // raised by the last rebuild's merge kernels
    SPH_HIP(hipMemsetAsync(s->slabCounts + 3, 0, sizeof(uint32_t), s->stream));
    sph_set_error("a halo message passed to the last sph_slab_rebuild was not sorted by global id");
    return SPH_ERR_INVALID;
  }
  if (h[7]) return slab_motion_error(s, h[7]);
  s->slabKept = (int)h[0];
  if ((int)h[1] > capRecords || (int)h[2] > capRecords) { sph_set_error("halo message overflow: %u / %u records, room for %d", h[1], h[2], capRecords); return SPH_ERR_SIZE; }
  return SPH_OK;
}

extern "C" int sph_slab_pack(sph_solver* s, void* msgDown, void* msgUp, int32_t capRecords, int32_t counts[3]) {
  ENTER(s);
  if (!s->hasSlab || !counts || capRecords < 0 || ((s->slab.hasLower && !msgDown) || (s->slab.hasUpper && !msgUp))) {
    sph_set_error("sph_slab_pack: slab not initialised or null message buffer"); return SPH_ERR_INVALID; }
  return slab_pack(s, (uint32_t*)msgDown, (uint32_t*)msgUp, capRecords, counts, nullptr, nullptr);
}

extern "C" int sph_slab_pack_framed(sph_solver* s, void* frameDown, void* frameUp, int32_t capRecords, int32_t counts[3]) {
  ENTER(s);
  if (!s->hasSlab || !counts || capRecords < 0 || ((s->slab.hasLower && !frameDown) || (s->slab.hasUpper && !frameUp))) {
    sph_set_error("sph_slab_pack_framed: slab not initialised or null frame buffer"); return SPH_ERR_INVALID; }
  uint32_t* fd = (uint32_t*)frameDown;
  uint32_t* fu = (uint32_t*)frameUp;
  return slab_pack(s, fd ? fd + 1 : nullptr, fu ? fu + 1 : nullptr, capRecords, counts, fd, fu);
}

// ---- overlapped step: sph_slab_step_begin enqueues everything and returns; sph_slab_step_messages blocks only until the
// messages are packed (the rest of the step is still running); sph_slab_rebuild then waits for the step itself.
extern "C" int sph_slab_step_begin(sph_solver* s, int iterationCount, void* frameDown, void* frameUp, int32_t capRecords) {
  (void)iterationCount;
  ENTER(s);
  if (!s->hasSlab || capRecords < 0 || (s->slab.hasLower && !frameDown) || (s->slab.hasUpper && !frameUp)) {
    sph_set_error("sph_slab_step_begin: slab not initialised or null frame buffer"); return SPH_ERR_INVALID; }
  if (s->slabStepPending) { sph_set_error("sph_slab_step_begin: the previous overlapped step was not rebuilt"); return SPH_ERR_ORDER; }
  StepTail tail{(uint32_t*)frameDown, (uint32_t*)frameUp, (int)capRecords};
  const int rc = enqueue_step(s, &tail);
  if (rc != SPH_OK) return rc;
  s->slabStepPending = true;
  s->slabKept = -1;
  s->slabCapRecords = (int)capRecords;
  return SPH_OK;
}

extern "C" int sph_slab_step_messages(sph_solver* s, int32_t counts[2]) {
  ENTER(s);
  if (!s->hasSlab || !s->slabStepPending || !counts) { sph_set_error("sph_slab_step_messages without sph_slab_step_begin"); return SPH_ERR_ORDER; }
  SPH_HIP(hipEventSynchronize(s->slabMsgEvent));
  counts[0] = (int32_t)s->slabHost[5]; counts[1] = (int32_t)s->slabHost[6];
  if (counts[0] > s->slabCapRecords || counts[1] > s->slabCapRecords) {
    sph_set_error("halo message overflow: %d / %d records, room for %d", counts[0], counts[1], s->slabCapRecords);
    return SPH_ERR_SIZE;
  }
  return SPH_OK;
}

extern "C" int sph_slab_rebuild(sph_solver* s, const void* recvDown, int32_t nDown, const void* recvUp, int32_t nUp) {
  ENTER(s);
  if (!s->hasSlab || nDown < 0 || nUp < 0 || (nDown && !recvDown) || (nUp && !recvUp)) { sph_set_error("sph_slab_rebuild: bad arguments"); return SPH_ERR_INVALID; }
  if (s->slabStepPending) {  // overlapped step: the kept count arrives with the end of the step
    SPH_HIP(hipStreamSynchronize(s->stream));
    s->slabStepPending = false;
    if (s->slabHost[3]) {
      SPH_HIP(hipMemsetAsync(s->slabCounts + 3, 0, sizeof(uint32_t), s->stream));
      sph_set_error("a halo message passed to the last sph_slab_rebuild was not sorted by global id");
      return SPH_ERR_INVALID;
    }
    if (s->slabHost[7]) return slab_motion_error(s, s->slabHost[7]);
    s->slabKept = (int)s->slabHost[8];
  }
  if (s->slabKept < 0) { sph_set_error("sph_slab_rebuild without a preceding sph_slab_pack"); return SPH_ERR_ORDER; }
  const int kept = s->slabKept;
  s->slabKept = -1;
  const long long total = (long long)kept + nDown + nUp;
  if (total > s->capacity || total <= 0) { sph_set_error("slab holds %lld particles after the exchange, capacity %d", total, s->capacity); return SPH_ERR_SIZE; }
  return sphk_slab_rebuild(s, (const uint32_t*)recvDown, nDown, (const uint32_t*)recvUp, nUp, kept);
}

// ---- the rebuild without a host round trip. The frames are what RCCL delivered: [payload words | payload]; the kept count is
// where the pack left it on the device. Everything is enqueued at once; the totals come back through pinned memory and
// slab_finish (called by sph_slab_rebuild_finish, or implicitly by the next entry point) sets the new particle count.
extern "C" int sph_slab_rebuild_framed(sph_solver* s, const void* frameDown, int32_t capDownRecords, const void* frameUp,
                                       int32_t capUpRecords) {
  ENTER(s);
  if (!s->hasSlab || capDownRecords < 0 || capUpRecords < 0 || (s->slab.hasLower && !frameDown) || (s->slab.hasUpper && !frameUp)) {
    sph_set_error("sph_slab_rebuild_framed: slab not initialised, negative capacity, or no frame from a neighbour that exists");
    return SPH_ERR_INVALID;
  }
  const uint32_t* keptPtr;
  if (s->slabStepPending) keptPtr = s->slabCounts + 8;       // overlapped step: the kept pass of sph_slab_step_begin
  else if (s->slabKept >= 0) keptPtr = s->slabCounts + 0;    // sph_slab_pack / sph_slab_pack_framed
  else { sph_set_error("sph_slab_rebuild_framed without a preceding pack"); return SPH_ERR_ORDER; }
  s->slabCapDown = frameDown ? capDownRecords : 0; s->slabCapUp = frameUp ? capUpRecords : 0;
  int rc = sphk_slab_rebuild_framed(s, (const uint32_t*)frameDown, s->slabCapDown, (const uint32_t*)frameUp, s->slabCapUp, keptPtr, s->slabCounts + 12);
  if (rc != SPH_OK) return rc;
  SPH_HIP(hipMemcpyAsync(s->slabHost + 12, s->slabCounts + 12, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s->stream));
  SPH_HIP(hipEventRecord(s->slabRebuildEvent, s->stream));
  s->slabRebuildPending = true;
  return SPH_OK;
}

// counts: kept, records from below, records from above, and 1 if NOTHING was merged because a frame announced more records than
// its buffer had room for (the caller fetches the missing part and rebuilds with sph_slab_rebuild; the state is untouched).
static int slab_finish(sph_solver* s, int32_t counts[4]) {
  SPH_HIP(hipEventSynchronize(s->slabRebuildEvent));
  s->slabRebuildPending = false;
  if (s->slabStepPending) {  // the flags that came back with the end of the overlapped step
    s->slabStepPending = false;
    if (s->slabHost[3]) {
      SPH_HIP(hipMemsetAsync(s->slabCounts + 3, 0, sizeof(uint32_t), s->stream));
      sph_set_error("a halo message passed to the last rebuild was not sorted by global id");
      return SPH_ERR_INVALID;
    }
    if (s->slabHost[7]) return slab_motion_error(s, s->slabHost[7]);
  }
  const int kept = (int)s->slabHost[12], nDown = (int)s->slabHost[13], nUp = (int)s->slabHost[14];
  const bool nothing = s->slabHost[15] != 0u;
  if (counts) { counts[0] = kept; counts[1] = nDown; counts[2] = nUp; counts[3] = nothing ? 1 : 0; }
  if (nothing) {
    if (nDown <= s->slabCapDown && nUp <= s->slabCapUp) {
      sph_set_error("slab holds %lld particles after the exchange, capacity %d", (long long)kept + nDown + nUp, s->capacity);
      return SPH_ERR_SIZE;
    }
    s->slabKept = kept;  // the frames were too short: sph_slab_rebuild with the complete messages finishes the job
    if (!counts) { sph_set_error("a halo frame announced %d / %d records, room for %d / %d", nDown, nUp, s->slabCapDown, s->slabCapUp); return SPH_ERR_SIZE; }
    return SPH_OK;
  }
  s->slabKept = -1;
  s->d.N = kept + nDown + nUp;
  s->progress = 0;
  return SPH_OK;
}

extern "C" int sph_slab_rebuild_finish(sph_solver* s, int32_t counts[4]) {
  ENTER_RAW(s);
  if (!counts) { sph_set_error("sph_slab_rebuild_finish: null argument"); return SPH_ERR_INVALID; }
  if (!s->slabRebuildPending) { sph_set_error("sph_slab_rebuild_finish without sph_slab_rebuild_framed"); return SPH_ERR_ORDER; }
  return slab_finish(s, counts);
}

extern "C" int sph_slab_liquid_signature(sph_solver* s, uint32_t* typeBits) {
  ENTER(s);
  if (!typeBits) return SPH_ERR_INVALID;
  *typeBits = s->liquidSig;
  return SPH_OK;
}

extern "C" int sph_slab_set_record_format(sph_solver* s, int32_t recordWords, uint32_t typeBits) {
  ENTER(s);
  if (recordWords != SPH_SLAB_RECORD_WORDS && recordWords != SPH_SLAB_COMPACT_WORDS) { sph_set_error("record words must be %d or %d", SPH_SLAB_RECORD_WORDS, SPH_SLAB_COMPACT_WORDS); return SPH_ERR_INVALID; }
  if (recordWords == SPH_SLAB_COMPACT_WORDS && (s->liquidSig == 0xffffffffu || (s->liquidSig != 0u && s->liquidSig != typeBits))) {
    sph_set_error("compact halo records need one common type word and velocity.w == 0 for every non-boundary particle");
    return SPH_ERR_INVALID;
  }
  s->slabRecWords = recordWords; s->slabTypeBits = typeBits;
  return SPH_OK;
}

extern "C" int sph_stream_wait_event(sph_solver* s, void* hipEvent) {
  ENTER(s);
  if (!hipEvent) return SPH_ERR_INVALID;
  SPH_HIP(hipStreamWaitEvent(s->stream, (hipEvent_t)hipEvent, 0));
  return SPH_OK;
}

extern "C" int sph_slab_read(sph_solver* s, float* position4, float* velocity4, uint32_t* globalIds, uint32_t* owned) {
  ENTER(s);
  if (!s->hasSlab) { sph_set_error("slab not initialised"); return SPH_ERR_INVALID; }
  const size_t n = (size_t)s->d.N;
  int rc = SPH_OK;
  if (position4) rc = d2h(s, position4, s->d.posOrig, sizeof(float4) * n);
  if (rc == SPH_OK && velocity4) rc = d2h(s, velocity4, s->d.velOrig, sizeof(float4) * n);
  if (rc == SPH_OK && globalIds) rc = d2h(s, globalIds, s->d.gid, sizeof(uint32_t) * n);
  if (rc == SPH_OK && owned) rc = d2h(s, owned, s->d.owned, sizeof(uint32_t) * n);
  return rc;
}
