// Internal declarations of libsphmi.so (gfx950 only). Data layout in HBM — see DESIGN.md §3.
//
//   orig order (API state)      posOrig  float4[N] (x,y,z,type)        velOrig float4[N] (vx,vy,vz,w)
//   sorted order (per step)     sortedPos float4[N] (x,y,z,type)       sortedVel float4[N]     predPos float[3N] (x,y,z)
//                               keys u32[N] (cell)  vals u32[N] (orig id)  backIndex u32[N] (orig -> sorted)
//                               rho f32[N]   rp float2[N] (rhoPred, pressure)   acc / accP float4[N]
//   neighbour map, tiled        nbr16 u16 (ids as offsets), nbrDist f32, nbrId i32 (rows nbr16 cannot hold):
//                               [tile = id/64][group = slot/4][lane = id%64][slot%4]
//                               -> one wave reads 4 slots of its 64 particles as ONE contiguous 512-B / 1-KiB transaction
//   grid                        cellStart u32[G+1]  (== gridCellIndexFixedUp: #particles with cell < c)
//
// float4 is kept for everything that other particles gather (one 16-B transaction per neighbour instead of
// three 4-B ones); pure per-particle streams (map, rho, pressure) are SoA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "sphmi.h"

#define SPH_BLOCK 256
#define SPH_TILE 64
#define SPH_MAXN 32
#define SPH_RSEG 30  // radius_segments, sphFluid.cl:116
#define SPH_DBG_WORDS 32  // diagnostic counters (SphDev::dbg)
#define SPH_SLAB_COUNT_WORDS 24
#define SPH_SORT_MAX_DIGITS 512   // radix digits per pass: 256 (8 bits) or 512 (9 bits)

struct SphDev {  // what the kernels see; passed by value
  int N, G;
  int gx, gy, gz;
  uint32_t cellMask;
  float h, cellSize, cellSizeInv, simScale, simScaleInv;
  float xmin, xmax, ymin, ymax, zmin, zmax;
  float r0, mass, rho0, dt, delta;
  float gravx, gravy, gravz;
  float surfTens;      // sphFluid.cl:662
  float massMu;        // (float)(mass*mu), sphFluid.cl:688
  float hs, hs2, hs6;  // hScaled, hScaled^2, hScaled^6 (float products as in sphFluid.cl:495-497)
  float posTimeStep;   // timeStep * simulationScaleInv (sphFluid.cl:928)
  float rho0delta;     // rho0*delta (sphFluid.cl:1168)
  double massWpoly6;   // ((double)mass)*Wpoly6Coefficient (sphFluid.cl:516)
  double massGradW;    // ((double)mass)*gradWspikyCoefficient (sphFluid.cl:1194)
  double del2W;        // del2WviscosityCoefficient
  double closeR;       // 0.5*(hScaled/2) as the double the comparison at sphFluid.cl:1166 uses
  float fastValueMin, fastD2Min, fastD2Max;  // operand bounds of k_pressure_force's short division / square-root path (sph_fast_bounds)
  float closeRf;       // the same test on floats: (double)r < closeR <=> r < closeRf (smallest float >= closeR)
  int rangeLo, rangeHi;  // a launch serves the sorted particles of cells [rangeLo, rangeHi): all of them ([0, G)) except in
                         // slab mode, where ghost layers that a stage's results are not needed on are skipped (sph_api.hip)
  int numElastic, elasticOffset, muscleCount, numMembranes;
  int hasElastic;      // 0: membrane kernels are no-ops and are folded into integrate
  // buffers
  float4 *posOrig, *velOrig, *membDelta;
  float4 *sortedPos, *sortedVel, *acc, *accP;
  float* predPos;    // predicted positions, packed (x, y, z): 3 floats per sorted particle
  uint32_t* elasticMask;  // bit k set: neighbour slot k holds an elastic particle (forces kernel -> membrane kernel)
  uint32_t* bndMask;  // bit k set: neighbour slot k holds a boundary particle (written by the forces kernel, read by integrate)
  float4* gatherRec; // 2 x ceil4(N) float4, groups of four particles [4 x (x,y,z,type)][4 x (v.xyz, rho)]: the forces kernel's neighbour gathers
  float2* rp;        // (rhoPred, pressure) per sorted particle: one 8-byte record, gathered per neighbour by the pressure-force kernel
  uint32_t *keys, *vals, *keysAlt, *valsAlt, *backIndex;
  uint32_t *cellStart, *cellStartRaw;
  uint32_t *gid, *owned;  // slab decomposition: global id and ownership flag per local particle (orig order)
  int32_t* nbrId;
  float* nbrDist;
  uint16_t* nbr16;    // the ids again as 16-bit offsets (see nbr16_index / SPH_N16_*): what the PCISPH gather kernels stream
  int32_t* nbrBase;   // per sorted particle: first sorted index of its z-neighbour cell, the base of the flagged offsets
  float* rho;
  float4* elastic;
  int32_t *membraneData, *pml;
  float* muscle;
  const float* binU;  // 64 floats: [0..31] d^2 < binU[j] <=> the candidate is counted in radial-histogram bins 0..j;
                      // [32..62] pass-1 radii r_thr(jb)^2; [63] the search kernel's filter radius^2 (sph_api.hip)
  uint32_t* dbg;  // SPH_DBG_WORDS diagnostic counters (neighbour-search fallbacks etc.), zeroed by sph_reset_stage_times
};

struct sph_solver {
  sph_config cfg;
  SphDev d;
  hipStream_t stream;
  bool ownStream;
  int sortBits;              // significant bits of the sort key
  int capacity;              // particles the buffers are sized for (>= d.N)
  int capTiles;              // ceil(capacity/64)
  sph_slab slab; bool hasSlab;
  uint32_t* slabCounts;      // device, SPH_SLAB_COUNT_WORDS words: [0..2] kept / down / up of sph_slab_pack, [3] unsorted-message flag,
                             // [4..6] and [8..10] the same triples of the overlapped step's message / kept passes, [7] owned
                             // particles that moved more than a layer, [12..15] totals of sph_slab_rebuild_framed (kept, from
                             // below, from above, 1 = nothing merged)
  uint32_t* slabHost;        // pinned host mirror of slabCounts (overlapped step)
  hipEvent_t slabMsgEvent;   // recorded when the messages of the overlapped step are packed
  bool slabStepPending;      // sph_slab_step_begin issued, rebuild not yet done
  int slabCapRecords;        // frame capacity given to sph_slab_step_begin
  int slabKept;              // host copy of the kept count of the last sph_slab_pack (-1: none pending)
  int slabRecWords;          // words per message record: 9 (full) or 7 (compact); 0 = 9
  uint32_t slabTypeBits;     // compact records: the type word every non-boundary particle carries
  uint32_t liquidSig;        // from sph_create: common position.w bits of the non-boundary particles with velocity.w == +0,
                             // 0 = there are none, 0xffffffff = not uniform
  bool slabRebuildPending;   // sph_slab_rebuild_framed issued: the particle count is still on its way to the host
  hipEvent_t slabRebuildEvent;
  int slabCapDown, slabCapUp;  // records the frames given to sph_slab_rebuild_framed had room for
  // radix-sort workspace
  uint32_t* blockHist;       // [SPH_SORT_MAX_DIGITS][maxSortBlocks] block histograms + SPH_SORT_MAX_DIGITS digit totals
  int maxSortBlocks;
  // stage progress for SPH_ERR_ORDER checks
  int progress;
  // stage timing
  bool timing;
  hipEvent_t evStart[SPH_ST_COUNT], evStop[SPH_ST_COUNT];
  struct Pending { int stage; hipEvent_t a, b; };
  Pending* pending; int numPending, capPending;
  double stageMs[SPH_ST_COUNT];
  int64_t stageLaunches[SPH_ST_COUNT];
  // host staging for exports
  void* hostScratch; size_t hostScratchBytes;
  // asynchronous position read-back (sph_read_position_async): a copy stream of its own, ordered against s->stream by events
  hipStream_t copyStream;
  hipEvent_t evReadReady, evCopyDone;  // positions final on s->stream / copy landed on the host
  bool copyPending;                    // a copy was issued and sph_read_position_wait has not run since
  float* copyUserDst;                  // where the caller wants the data
  void* copyStage; size_t copyStageBytes;  // pinned staging, only for destinations that cannot be page-locked in place
  bool copyViaStage;
  struct HostReg { void* p; size_t bytes; };
  HostReg hostRegs[8]; int numHostRegs;    // caller buffers page-locked in place by hipHostRegister (released by sph_destroy)
  uint32_t* pinnedFlags;               // pinned: [0] copy of dbg[6] taken with the last asynchronous read-back
  uint64_t blownUp;                    // sticky: non-finite coordinates seen so far (check_finite_state)
};

// Called by every launcher whose kernel WRITES posOrig (integrate, membranes finalize, slab rebuild): makes s->stream wait for
// an asynchronous position read-back that is still in flight.
int sph_guard_position_write(sph_solver* s);

void sph_set_error(const char* fmt, ...);

#define SPH_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess) {                                                             \
      sph_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SPH_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

static inline int sph_blocks(int n, int per = SPH_BLOCK) { return (n + per - 1) / per; }

// SphDev for a launch restricted to the owned layers plus `ghostDepth` ghost layers per side (slab mode; otherwise and
// for ghostDepth < 0 the whole particle set).
SphDev sph_ranged(const sph_solver* s, int ghostDepth);

#ifdef __HIPCC__
// id of the linear-th particle of the launch's range; false past its end
__device__ __forceinline__ bool sph_range_id(const SphDev& d, int linear, int& id) {
  const int begin = (int)d.cellStart[d.rangeLo], end = (int)d.cellStart[d.rangeHi];
  id = begin + linear;
  return id < end;
}
#endif


// index of (sorted particle id, slot) in the tiled neighbour map
__host__ __device__ static inline size_t nbr_index(int id, int slot) {
  return ((((size_t)(id >> 6) * 8 + (size_t)(slot >> 2)) * 64 + (size_t)(id & 63)) << 2) + (size_t)(slot & 3);
}

// The neighbour ids a second time, 2 bytes each, in the layout of the other two maps ([tile = id/64][group = slot/4][lane = id%64]
// [slot%4]: same element index, so findNeighbors computes one address per entry): a wave reads 4 slots of its 64 particles as
// one contiguous 512-byte transaction and a particle's 32 ids are 64 bytes instead of 128. The three PCISPH
// gather kernels are bound by the bytes they stream per particle, and the id rows were 40-85 % of those.
//   entry = 0xFFFF                       empty slot
//   entry = flag << 15 | (j - base + SPH_N16_BIAS)   with base = (flag ? nbrBase[id] : id); 0 <= low 15 bits <= SPH_N16_MAX
//   entry 0 of a row = SPH_N16_WIDE      the row could not be encoded (an offset out of range, or the particle took the exact
//                                        walk): readers take the 32-bit row of nbrId instead
// Neighbours lie in the particle's own x-row of cells, the adjacent y-row (both within (cells per row + 2) x occupancy of the
// particle itself) or the same two rows one z layer away (within that distance of the z-neighbour cell's first particle).
#define SPH_N16_BIAS 16384
#define SPH_N16_MAX 0x7FFD
#define SPH_N16_EMPTY 0xFFFFu
#define SPH_N16_WIDE 0xFFFEu
// The sorted index in slot `slot` of particle `id` (-1: empty) from the arrays (any mix of host copies or device pointers):
// findNeighbors writes the 32-bit row only where the 16-bit one could not be written.
__host__ __device__ static inline int nbr_decode(const uint16_t* n16, const int32_t* nbase, const int32_t* n32, int id, int slot) {
  const size_t row0 = nbr_index(id, 0), at = nbr_index(id, slot);
  if (n16[row0] == SPH_N16_WIDE) return n32[at];
  const uint32_t e = n16[at];
  if (e == SPH_N16_EMPTY) return -1;
  return ((e & 0x8000u) ? nbase[id] : id) + (int)(e & 0x7fffu) - SPH_N16_BIAS;
}

// ---- launchers (each enqueues on s->stream and returns SPH_OK / SPH_ERR_HIP) ----
// sph_sort.hip
int sphk_hash(sph_solver* s);
int sphk_sort(sph_solver* s);
int sphk_sort_pairs(sph_solver* s, int n, int bits);  // stable LSD sort of (keys, vals)[0..n) by the low `bits` of keys
int sph_sort_passes(int bits);                         // radix passes sphk_sort_pairs takes for `bits` key bits (8- or 9-bit digits)
int sphk_step_sort_bits(const sph_solver* s, bool* compact);         // key bits the fused step sorts (compacted keys or the real ones)
int sphk_hash_for_step(sph_solver* s, int* sortBits, bool* compact);  // fused step: hash with compacted keys where that saves a pass
int sphk_sort_post_rekey(sph_solver* s);               // the gather after a sort of compacted keys (puts the real cell ids back)
int sphk_sort_post(sph_solver* s);        // gather + backIndex (K3)
int sphk_index_raw(sph_solver* s);        // K4 table with -1 for empty cells
int sphk_index_fixed(sph_solver* s);      // H2 table (cellStart)
int sphk_sort_post_and_index(sph_solver* s);  // fused K3 + K4 + H2
int sphk_hash_sort_post_slab(sph_solver* s);  // slab mode: K2 + sort + K3 with compacted sort keys (2 radix passes)
// sph_neighbors.hip
int sphk_clear_neighbors(sph_solver* s);
int sphk_find_neighbors(sph_solver* s, int ghostDepth = -1);
// sph_pcisph.hip
int sphk_density(sph_solver* s, int ghostDepth = -1);
int sphk_forces(sph_solver* s, bool fusePredict, int ghostDepth = -1);
int sphk_ghost_init(sph_solver* s);  // what K7 does besides the acceleration, for every particle (slab mode)
int sphk_predict_positions(sph_solver* s);
int sphk_predict_density(sph_solver* s, bool fuseCorrect, int ghostDepth = -1, bool first = false);  // first: fused step, iteration 0
int sphk_correct_pressure(sph_solver* s);
int sphk_pressure_force(sph_solver* s, int fuse, int ghostDepth = -1);  // 0 none, 1 + predictPositions, 2 + integrate
int sphk_integrate(sph_solver* s);
// sph_slab.hip
// Which part of the pack a launch does. ALL: kept set + both messages (sph_slab_pack). The overlapped step packs the MESSAGES
// as soon as the particles near the cuts (CELL ranges [a0,a1) and [b0,b1) of the sorted order) are integrated, and the KEPT set
// at the end.
enum { SLAB_PART_ALL = 0, SLAB_PART_MESSAGES = 1, SLAB_PART_KEPT = 2 };
struct SlabPart { int mode, a0, a1, b0, b1; };
int sphk_slab_pack(sph_solver* s, uint32_t* msgDown, uint32_t* msgUp, int capRecords, uint32_t* headDown = nullptr,
                   uint32_t* headUp = nullptr,  // head*: where to store the payload word count of each message (framed mode)
                   SlabPart part = SlabPart{SLAB_PART_ALL, 0, 0, 0, 0}, uint32_t* counts = nullptr);  // counts: 3 device words (default slabCounts)
SphDev sph_ranged_layers(const sph_solver* s, long long loLayer, long long hiLayer);  // launch restricted to cell layers [lo, hi)
int sphk_pressure_force_layers(sph_solver* s, int fuse, long long loLayer, long long hiLayer);
int sphk_slab_rebuild(sph_solver* s, const uint32_t* recvDown, int nDown, const uint32_t* recvUp, int nUp, int kept);  // 3-way merge
int sphk_slab_rebuild_framed(sph_solver* s, const uint32_t* frameDown, int capDown, const uint32_t* frameUp, int capUp,
                             const uint32_t* keptPtr, uint32_t* totals);  // every length read on the device; d.N is left alone
int sphk_slab_sort_rebuild(sph_solver* s, int total);  // staging area in any order -> local set sorted by global id
// sph_elastic.hip
int sphk_elastic(sph_solver* s);
int sphk_clear_membranes(sph_solver* s);
int sphk_membranes(sph_solver* s);
int sphk_membranes_finalize(sph_solver* s);
