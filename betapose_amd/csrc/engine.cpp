// Plan construction + execution for the detector (YOLOv3 from a Darknet cfg) and the
// key-point detector (FastPose = SE-ResNet-101 + DUC).  Host side only: parsing,
// BN folding, filter packing, buffer layout, launch order.  All arithmetic of the hot
// path runs in the HIP kernels of conv_igemm.hip / aux_kernels.hip.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace bp {

// ------------------------------------------------------------------ arena
Arena::~Arena() {
    for (void* p : ptrs_) (void)hipFree(p);
}
void* Arena::alloc_bytes(size_t bytes) {
    void* p = nullptr;
    bytes = (bytes + 255) & ~size_t(255);
    BP_HIP(hipMalloc(&p, bytes));
    ptrs_.push_back(p);
    total_ += bytes;
    return p;
}
float* Arena::alloc(size_t n) { return static_cast<float*>(alloc_bytes(n * sizeof(float))); }

// ------------------------------------------------------------------ Net
Tensor Net::new_tensor(int H, int W, int C) {
    Tensor t;
    t.H = H; t.W = W; t.C = C; t.ld = C;
    t.p = arena_.alloc((size_t)max_batch_ * H * W * C);
    ActAlloc a; a.base = t.p; a.elems = (size_t)max_batch_ * H * W * C;
    acts_.push_back(a);
    return t;
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }
static int env_int_early(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

float* Net::upload_weights(const float* host, size_t count) {
    if (reuse_) {
        BP_CHECK(store_cursor_ < store_->ptrs.size(), "weight store exhausted (clone of a different network?)");
        return store_->ptrs[store_cursor_++];
    }
    float* d = store_->arena.alloc(count);
    BP_HIP(hipMemcpy(d, host, count * sizeof(float), hipMemcpyHostToDevice));
    store_->ptrs.push_back(d);
    return d;
}

// Split-K slice counts measured on MI355X for every batch-1 conv shape of the two networks, one kernel at a time
// (tools/tune_conv.py, 64x64 tile): {M, CoutPad, K-chunks, slices}.  Other shapes use the heuristic below.
struct SplitEntry { int M, CoutPad, nchunks, splits; };
static const SplitEntry kSplitTable[] = {
    {    80,   512,   64,  6},
    {    80,   512,  144, 12},
    {    80,  2048,   16,  3},
    {    80,  2048,   32,  3},
    {   169,    64,   32, 10},
    {   169,   256,   16,  5},
    {   169,   512,   32,  5},
    {   169,  1024,  144,  5},
    {   320,   256,   32,  5},
    {   320,   256,   72,  8},
    {   320,   512,   32,  5},
    {   320,  1024,    8,  1},
    {   320,  1024,   16,  3},
    {   320,  1024,  144,  6},
    {   676,    64,   16,  4},
    {   676,   128,    8,  3},
    {   676,   256,   16,  3},
    {   676,   256,   24,  5},
    {   676,   512,   72,  5},
    {  1280,   128,   16,  3},
    {  1280,   128,   36,  5},
    {  1280,   256,   16,  3},
    {  1280,   512,    4,  1},
    {  1280,   512,    8,  1},
    {  1280,   512,   72,  3},
    {  2704,    64,    8,  1},
    {  2704,   128,    8,  1},
    {  2704,   128,   12,  2},
    {  2704,   256,   36,  4},
    {  5120,    64,    2,  1},
    {  5120,    64,    8,  1},
    {  5120,    64,   18,  3},
    {  5120,    64,   36,  3},
    {  5120,   128,    8,  1},
    {  5120,   256,    2,  1},
    { 10816,    64,    4,  1},
    { 10816,   128,   18,  3},
    { 43264,    64,    2,  1},
    { 43264,    64,    9,  1},
};

// 16-bit precision modes: {M, CoutPad, K-chunks} -> {tile, slices}, measured one kernel at a time over both kernel
// families (tools/tune_conv.py [--f16 | --kg], profiles/r02_tune_{b3,f16,b3_kernels}.txt: at batch 1 the 64x64-block
// kernels of conv_igemm.hip win every shape of the two networks, in the bf16x3 mode its filters-direct variant; the
// slice counts are those the whole pipeline runs fastest with, which are higher than a kernel timed alone prefers);
// other shapes use the heuristic in choose_h16.
// COVERAGE: the rows below are the conv shapes of the two networks at BATCH 1 (M = OH x OW of one 416x416 frame / one 320x256 crop);
// the conv_pl tables further down also carry batch 28 (BASELINE configs[2]).  Any other batch size or input resolution takes the
// heuristics in choose_h16 / choose_pl, which are measured at batch 2, 4 and 28 only (tools/batch_check.sh, profiles/r04_batched.txt).
struct PlanEntry { int M, CoutPad, nchunks, tile, splits; };
static const PlanEntry kPlanB3[] = {
    // (round 4, TILE_BD_K2 rows: the filters-direct tile with two K groups inside an eight-wave block and about half the K slices
    // between blocks -- alone it is no faster than the four-wave tile on any shape (tools/bench_bdk2.py, profiles/r04_bdk2_kernels.txt),
    // in the pipeline the plan with these rows is +2-2.8 % on three boxes (tools/plans/bdk2*.txt, profiles/r04_ab_bdk2.txt): fewer
    // blocks and slabs for the same work)
    // round 4, conv_halo.hip: the 3x3 / stride-1 layers on the tap-resident halo tile -- rows apply where conv_halo_eligible()
    // holds (a stride-2 layer of the same {M, CoutPad, K-chunks} falls through to its filters-direct row below).  Slice counts from
    // A/B runs of the whole pipeline, four frames in flight (profiles/r04_halo_ab.txt)
    // (first the 64x128 tile with 8 / 6 / 4 / 3 / 2 slices: +4-5 %; then the 64x64 tile with two K groups inside the block and half the
    // slices -- none at 52x52: a further +1.2-2 %, tools/plans/k2*.txt)
    {   169,  1024,  144, TILE_HALO64K2,  4},
    {   320,  1024,  144, TILE_HALO64K2,  4},
    {   676,   512,   72, TILE_HALO64K2,  2},
    {  1280,   512,   72, TILE_HALO64K2,  2},
    {  2704,   256,   36, TILE_HALO64K2,  1},
    {    80,   512,   64, TILE_BD_K2,  4},
    {    80,   512,  144, TILE_BD_K2,  6},
    {    80,  2048,   16, TILE_BD_K2,  2},
    {    80,  2048,   32, TILE_BD_K2,  2},
    {   169,    64,   32, TILE_BD_K2,  6},
    {   169,   256,   16, TILE_BD_K2,  3},
    {   169,   512,   32, TILE_BD_K2,  6},
    {   169,  1024,  144, TILE_64x64_BD,  5},
    {   320,   256,   32, TILE_BD_K2,  6},
    {   320,   256,   72, TILE_BD_K2,  6},
    {   320,   512,   32, TILE_BD_K2,  3},
    {   320,  1024,    8, TILE_BD_K2,  1},
    {   320,  1024,   16, TILE_BD_K2,  2},
    {   320,  1024,  144, TILE_64x64_BD,  6},
    {   676,    64,   16, TILE_BD_K2,  3},
    {   676,   128,    8, TILE_BD_K2,  1},
    {   676,   256,   16, TILE_BD_K2,  3},
    {   676,   256,   24, TILE_BD_K2,  3},
    {   676,   512,   72, TILE_64x64_BD,  5},
    {  1280,   128,   16, TILE_BD_K2,  3},
    {  1280,   128,   36, TILE_BD_K2,  3},
    {  1280,   256,   16, TILE_BD_K2,  2},
    {  1280,   512,    4, TILE_64x64_BD,  1},
    {  1280,   512,    8, TILE_BD_K2,  1},
    {  1280,   512,   72, TILE_64x64_BD,  3},
    {  2704,    64,    8, TILE_BD_K2,  1},
    {  2704,   128,    8, TILE_BD_K2,  1},
    {  2704,   128,   12, TILE_BD_K2,  1},
    {  2704,   256,   36, TILE_64x64_BD,  2},   // in the pipeline: 2 slices 933, 3: 929, 4: 921, 5+: 910 frames/s (tools/tune_splits_insitu.py)
    {  5120,    64,    2, TILE_64x64_BD,  1},
    {  5120,    64,    8, TILE_BD_K2,  1},
    {  5120,    64,   18, TILE_BD_K2,  2},
    {  5120,    64,   36, TILE_BD_K2,  2},
    {  5120,   128,    8, TILE_BD_K2,  1},
    {  5120,   256,    2, TILE_64x64_BD,  1},
    { 10816,    64,    4, TILE_BD_K2,  1},
    { 10816,   128,   18, TILE_BD_K2,  1},
    { 43264,    64,    2, TILE_64x64_BD,  1},
    { 43264,    64,    9, TILE_64x64_BD,  1},
    {0, 0, 0, 0, 0},
};
// The lone-frame latency mode (Net::set_prefetch) keeps the four-wave filters-direct tile on the rows the table above gives to TILE_BD_K2:
// its XCD-local hand-off and filter prefetch blocks exist for that tile (conv_home_layout, conv_prefetch_of), and one frame at a time
// they are worth more than the K groups (2.46 against 2.56 ms per frame).  Slice counts: the round-3 table.
static const PlanEntry kPlanB3Lone[] = {
    {    80,   512,   64, TILE_64x64_BD,  6},
    {    80,   512,  144, TILE_64x64_BD, 10},
    {    80,  2048,   16, TILE_64x64_BD,  3},
    {    80,  2048,   32, TILE_64x64_BD,  3},
    {   169,    64,   32, TILE_64x64_BD, 10},
    {   169,   256,   16, TILE_64x64_BD,  5},
    {   169,   512,   32, TILE_64x64_BD,  5},
    {   320,   256,   32, TILE_64x64_BD,  5},
    {   320,   256,   72, TILE_64x64_BD,  8},
    {   320,   512,   32, TILE_64x64_BD,  5},
    {   320,  1024,    8, TILE_64x64_BD,  1},
    {   320,  1024,   16, TILE_64x64_BD,  3},
    {   676,    64,   16, TILE_64x64_BD,  5},
    {   676,   128,    8, TILE_64x64_BD,  1},
    {   676,   256,   16, TILE_64x64_BD,  3},
    {   676,   256,   24, TILE_64x64_BD,  5},
    {  1280,   128,   16, TILE_64x64_BD,  3},
    {  1280,   128,   36, TILE_64x64_BD,  5},
    {  1280,   256,   16, TILE_64x64_BD,  3},
    {  1280,   512,    8, TILE_64x64_BD,  1},
    {  2704,    64,    8, TILE_64x64_BD,  1},
    {  2704,   128,    8, TILE_64x64_BD,  1},
    {  2704,   128,   12, TILE_64x64_BD,  1},
    {  5120,    64,    8, TILE_64x64_BD,  1},
    {  5120,    64,   18, TILE_64x64_BD,  3},
    {  5120,    64,   36, TILE_64x64_BD,  3},
    {  5120,   128,    8, TILE_64x64_BD,  1},
    { 10816,    64,    4, TILE_64x64_BD,  1},
    { 10816,   128,   18, TILE_64x64_BD,  2},
    {0, 0, 0, 0, 0},
};
#ifdef BP_EXPERIMENTAL
// (fp16, batch 1: the filters-direct variant wins 40 of 47 shapes alone by 4.5 % in the sum, profiles/r02_tune_f16.txt, and
// LOSES in the pipeline -- 1 329 against 1 381 frames/s, A/B on one box -- so the batch-1 rows stay on the staged kernel)
static const PlanEntry kPlanF16[] = {
    {    80,   512,   64, 0,  5},
    {    80,   512,  144, 0,  6},
    {    80,  2048,   16, 0,  1},
    {    80,  2048,   32, 0,  3},
    {   169,    64,   32, 0,  4},
    {   169,   256,   16, 0,  1},
    {   169,   512,   32, 0,  3},
    {   169,  1024,  144, 0,  5},
    {   320,   256,   32, 0,  3},
    {   320,   256,   72, 0,  5},
    {   320,   512,   32, 0,  3},
    {   320,  1024,    8, 0,  1},
    {   320,  1024,   16, 0,  1},
    {   320,  1024,  144, 0,  5},
    {   676,    64,   16, 0,  1},
    {   676,   128,    8, 0,  1},
    {   676,   256,   16, 0,  1},
    {   676,   256,   24, 0,  1},
    {   676,   512,   72, 0,  5},
    {  1280,   128,   16, 0,  1},
    {  1280,   128,   36, 0,  3},
    {  1280,   256,   16, 0,  1},
    {  1280,   512,    4, 0,  1},
    {  1280,   512,    8, 0,  1},
    {  1280,   512,   72, 0,  3},
    {  2704,    64,    8, 0,  1},
    {  2704,   128,    8, 0,  1},
    {  2704,   128,   12, 0,  1},
    {  2704,   256,   36, 0,  1},
    {  5120,    64,    2, 0,  1},
    {  5120,    64,    8, 0,  1},
    {  5120,    64,   18, 0,  1},
    {  5120,    64,   36, 0,  3},
    {  5120,   128,    8, 0,  1},
    {  5120,   256,    2, 0,  1},
    { 10816,    64,    4, 0,  1},
    { 10816,   128,   18, 0,  1},
    { 43264,    64,    2, 0,  1},
    { 43264,    64,    9, 0,  1},
    // batch 28 (BASELINE configs[2]; profiles/r02_tune_f16_batch28.txt, with the 128x64 block and the filters-direct kernel among the candidates)
    {  2240,   512,   64, 1,  1},
    {  2240,   512,  144, 1,  1},
    {  2240,  2048,   16, 0,  1},
    {  2240,  2048,   32, 0,  1},
    {  4732,    64,   32, 12,  1},
    {  4732,   256,   16, 1,  1},
    {  4732,   512,   32, 12,  1},
    {  4732,  1024,  144, 1,  1},
    {  8960,   256,   32, 6,  1},
    {  8960,   256,   72, 6,  1},
    {  8960,   512,   32, 0,  1},
    {  8960,  1024,    8, 0,  1},
    {  8960,  1024,   16, 0,  1},
    {  8960,  1024,  144, 1,  1},
    { 18928,    64,   16, 0,  1},
    { 18928,   128,    8, 6,  1},
    { 18928,   256,   16, 0,  1},
    { 18928,   256,   24, 0,  1},
    { 18928,   512,   72, 6,  1},
    { 35840,   128,   16, 0,  1},
    { 35840,   128,   36, 0,  1},
    { 35840,   256,   16, 6,  1},
    { 35840,   512,    4, 0,  1},
    { 35840,   512,    8, 0,  1},
    { 35840,   512,   72, 6,  1},
    { 75712,    64,    8, 0,  1},
    { 75712,   128,    8, 0,  1},
    { 75712,   128,   12, 6,  1},
    { 75712,   256,   36, 6,  1},
    {143360,    64,    2, 0,  1},
    {143360,    64,    8, 0,  1},
    {143360,    64,   18, 12,  1},
    {143360,    64,   36, 12,  1},
    {143360,   128,    8, 6,  1},
    {143360,   256,    2, 0,  1},
    {302848,    64,    4, 0,  1},
    {302848,   128,   18, 6,  1},
    {1211392,    64,    2, 12,  1},
    {1211392,    64,    9, 1,  1},
    {0, 0, 0, 0, 0},
};
#endif   // BP_EXPERIMENTAL

// experiment hook (tools only): BP_PLAN_FILE names a text file of "M CoutPad nchunks tile splits" lines that take
// precedence over the built-in bf16x3 table, so a tuning sweep can be tried in the whole pipeline without a rebuild
static const std::vector<PlanEntry>& plan_file_entries() {
    static const std::vector<PlanEntry> entries = [] {
        std::vector<PlanEntry> v;
        if (const char* path = std::getenv("BP_PLAN_FILE")) {
            if (FILE* f = std::fopen(path, "r")) {
                PlanEntry e;
                while (std::fscanf(f, "%d %d %d %d %d", &e.M, &e.CoutPad, &e.nchunks, &e.tile, &e.splits) == 5) v.push_back(e);
                std::fclose(f);
            }
        }
        return v;
    }();
    return entries;
}

// conv_pl.hip, bf16x3 mode, batch 1: {M, CoutPad, K-chunks} -> {tile, slices}, every conv shape of the two networks timed alone with the
// epilogue the networks run (residual + operand planes; tools/tune_conv.py --pl, profiles/r03_tune_pl_b3.txt)
static const PlanEntry kPlanPL3[] = {
    {    80,   512,   64, TILE_PL64,  6},
    {    80,   512,  144, TILE_PL64, 10},
    {    80,  2048,   16, TILE_PL64,  3},
    {    80,  2048,   32, TILE_PL64,  3},
    {   169,    64,   32, TILE_PL64, 10},
    {   169,   256,   16, TILE_PL64,  4},
    {   169,   512,   32, TILE_PL64,  4},
    {   169,  1024,  144, TILE_PL64,  5},
    {   320,   256,   32, TILE_PL64,  4},
    {   320,   256,   72, TILE_PL64,  5},
    {   320,   512,   32, TILE_PL64,  3},
    {   320,  1024,    8, TILE_PL64,  1},
    {   320,  1024,   16, TILE_PL64,  1},
    {   320,  1024,  144, TILE_PL64,  3},
    {   676,    64,   16, TILE_PL64,  5},
    {   676,   128,    8, TILE_PL64,  1},
    {   676,   256,   16, TILE_PL64,  3},
    {   676,   256,   24, TILE_PL64,  3},
    {   676,   512,   72, TILE_PL64,  2},
    {  1280,   128,   16, TILE_PL64,  3},
    {  1280,   128,   36, TILE_PL64,  4},
    {  1280,   256,   16, TILE_PL64,  1},
    {  1280,   512,    4, TILE_PL64,  1},
    {  1280,   512,    8, TILE_PL64,  1},
    {  1280,   512,   72, TILE_PL64,  3},
    {  2704,    64,    8, TILE_PL64,  1},
    {  2704,   128,    8, TILE_PL64,  1},
    {  2704,   128,   12, TILE_PL64,  1},
    {  2704,   256,   36, TILE_PL64,  1},
    {  5120,    64,    2, TILE_PL64,  1},
    {  5120,    64,    8, TILE_PL64,  1},
    {  5120,    64,   18, TILE_PL64,  1},
    {  5120,    64,   36, TILE_PL64,  2},
    {  5120,   128,    8, TILE_PL64,  1},
    {  5120,   256,    2, TILE_PL64,  1},
    { 10816,    64,    4, TILE_PL64,  1},
    { 10816,   128,   18, TILE_PL64,  1},
    { 43264,    64,    2, TILE_PL64,  1},
    { 43264,    64,    9, TILE_PL64,  1},
    {0, 0, 0, 0, 0},
};
// ... fp16 mode (tools/tune_conv.py --pl --f16, profiles/r03_tune_pl_f16.txt)
static const PlanEntry kPlanPL1[] = {
    {    80,   512,   64, TILE_PL64,  4},
    {    80,   512,  144, TILE_PL64,  6},
    {    80,  2048,   16, TILE_PL64,  1},
    {    80,  2048,   32, TILE_PL64,  1},
    {   169,    64,   32, TILE_PL64,  4},
    {   169,   256,   16, TILE_PL64,  3},
    {   169,   512,   32, TILE_PL64,  4},
    {   169,  1024,  144, TILE_PL64,  4},
    {   320,   256,   32, TILE_PL64,  4},
    {   320,   256,   72, TILE_PL64,  4},
    {   320,   512,   32, TILE_PL64,  3},
    {   320,  1024,    8, TILE_PL64,  1},
    {   320,  1024,   16, TILE_PL64,  1},
    {   320,  1024,  144, TILE_PL64,  3},
    {   676,    64,   16, TILE_PL64,  1},
    {   676,   128,    8, TILE_PL64,  1},
    {   676,   256,   16, TILE_PL64,  1},
    {   676,   256,   24, TILE_PL64,  1},
    {   676,   512,   72, TILE_PL64,  2},
    {  1280,   128,   16, TILE_PL64,  1},
    {  1280,   128,   36, TILE_PL64,  3},
    {  1280,   256,   16, TILE_PL64,  1},
    {  1280,   512,    4, TILE_PL64,  1},
    {  1280,   512,    8, TILE_PL64,  1},
    {  1280,   512,   72, TILE_PL64,  3},
    {  2704,    64,    8, TILE_PL64,  1},
    {  2704,   128,    8, TILE_PL64,  1},
    {  2704,   128,   12, TILE_PL64,  1},
    {  2704,   256,   36, TILE_PL64,  1},
    {  5120,    64,    2, TILE_PL64,  1},
    {  5120,    64,    8, TILE_PL64,  1},
    {  5120,    64,   18, TILE_PL64,  1},
    {  5120,    64,   36, TILE_PL64,  1},
    {  5120,   128,    8, TILE_PL64,  1},
    {  5120,   256,    2, TILE_PL64,  1},
    { 10816,    64,    4, TILE_PL64,  1},
    { 10816,   128,   18, TILE_PL64,  1},
    { 43264,    64,    2, TILE_PL64,  1},
    { 43264,    64,    9, TILE_PL64,  1},
    // batch 28 (BASELINE configs[2]; tools/tune_conv.py --pl --f16 --big --batch 28, profiles/r03_tune_pl_f16_batch28.txt)
    // round 4: the 3x3 / stride-1 layers on the halo form of the 128x128 tile (TILE_PLH128; tools/bench_plh.py, profiles/r04_plh_kernels.txt:
    // 7-10 % faster alone than the best all-DMA tile, +1.7-2.6 % on configs[2]).  A row is taken only by layers the tile can run
    // (choose_pl checks conv_plh_eligible): the stride-2 layers that share a row's key fall through to the row below it
    {   4732,  1024,  144, TILE_PLH128,  1},
    {   8960,  1024,  144, TILE_PLH128,  1},   // DUC1 / DUC2 (round 5: their PixelShuffle stores take the staged epilogue in conv_pl.hip too)
    {  35840,   512,   72, TILE_PLH128,  1},
    {   8960,   256,   72, TILE_PLH128,  1},
    {  18928,   512,   72, TILE_PLH128,  1},
    {  35840,   128,   36, TILE_PLH128,  1},
    {  75712,   256,   36, TILE_PLH128,  1},
    { 302848,   128,   18, TILE_PLH128,  1},   // the 104x104 layers (round 5: 384 halo rows); BP_PLH_W63=1 keeps them on the 256x128 all-DMA tile (A/B runs)
    {   2240,   512,   64, TILE_PL64,  1},
    {   2240,   512,  144, TILE_PL64,  1},
    {   2240,  2048,   16, TILE_PL64,  1},
    {   2240,  2048,   32, TILE_PL128,  1},
    {   4732,    64,   32, TILE_PL64,  1},
    {   4732,   256,   16, TILE_PL128x64,  1},
    {   4732,   512,   32, TILE_PL64,  1},
    {   4732,  1024,  144, TILE_PL128,  1},
    {   8960,   256,   32, TILE_PL64,  1},
    {   8960,   256,   72, TILE_PL64,  1},
    {   8960,   512,   32, TILE_PL64,  1},
    {   8960,  1024,    8, TILE_PL64,  1},
    {   8960,  1024,   16, TILE_PL64,  1},
    {   8960,  1024,  144, TILE_PL256x128,  1},
    {  18928,    64,   16, TILE_PL64,  1},
    {  18928,   128,    8, TILE_PL64,  1},
    {  18928,   256,   16, TILE_PL64,  1},
    {  18928,   256,   24, TILE_PL64,  1},
    {  18928,   512,   72, TILE_PL256x128,  1},
    {  35840,   128,   16, TILE_PL64,  1},
    {  35840,   128,   36, TILE_PL64,  1},
    {  35840,   256,   16, TILE_PL64,  1},
    {  35840,   512,    4, TILE_PL64,  1},
    {  35840,   512,    8, TILE_PL64,  1},
    {  35840,   512,   72, TILE_PL256x128,  1},
    {  75712,    64,    8, TILE_PL64,  1},
    {  75712,   128,    8, TILE_PL64,  1},
    {  75712,   128,   12, TILE_PL256x128,  1},
    {  75712,   256,   36, TILE_PL128,  1},
    { 143360,    64,    2, TILE_PL64,  1},
    { 143360,    64,    8, TILE_PL64,  1},
    { 143360,    64,   18, TILE_PL64,  1},
    { 143360,    64,   36, TILE_PL64,  1},
    { 143360,   128,    8, TILE_PL64,  1},
    { 143360,   256,    2, TILE_PL64,  1},
    { 302848,    64,    4, TILE_PL64,  1},
    { 302848,   128,   18, TILE_PL256x128,  1},
    {1211392,    64,    2, TILE_PL64,  1},
    {1211392,    64,    9, TILE_PL64,  1},
    {0, 0, 0, 0, 0},
};

// the halo plane tile runs the layer; BP_PLH_W63=1: maps up to 63 wide only, as before round 5 (A/B runs)
static bool plh_ok(const ConvParams& c) {
    static const bool w63 = std::getenv("BP_PLH_W63") != nullptr;
    return conv_plh_eligible(c) && (c.W <= 63 || !w63);
}

// conv_pl.hip (operand planes + LDS-DMA): which block tile, how many K slices
static void choose_pl(const ConvParams& c, long long M, int mode, int sk_max, int* tile, int* splits) {
    for (const PlanEntry& e : plan_file_entries())
        if (e.M == (int)M && e.CoutPad == c.CoutPad && e.nchunks == c.nchunks && conv_tile_is_pl(e.tile) &&
            (!conv_tile_is_plh(e.tile) || plh_ok(c)) &&
            (e.tile != TILE_S1 || (e.splits == 1 && conv_s1_eligible(c, M))) &&
            (e.tile != TILE_P3 || (e.splits == 1 && conv_p3_eligible(c, M)))) { *tile = e.tile; *splits = e.splits; return; }   // (a plan-file row naming the streaming 1x1 kernel for a layer it cannot run is ignored, not a failed launch)
    for (const PlanEntry* e = (mode == PREC_F16 ? kPlanPL1 : kPlanPL3); e->M != 0; ++e)   // tables end with a zero row
        if (e->M == (int)M && e->CoutPad == c.CoutPad && e->nchunks == c.nchunks &&
            (!conv_tile_is_plh(e->tile) || plh_ok(c))) { *tile = e->tile; *splits = e->splits; return; }     // (a 3x3 row also matches stride-2 / 1x1 layers of the same K)
    // other shapes (batched runs): the 128x128 block once its grid covers the chip (half the operand bytes per FLOP of the
    // 64x64 block: profiles/r03_bench_pl_batch28.txt), else the 64x64 block with enough K slices to fill it
    // (short K loops -- the 1x1 layers of the bottlenecks -- stay on the 64x64 block even then: a 128x128 block runs one per
    // CU and its prologue / epilogue are not covered by a neighbour's K loop; profiles/r03_tune_pl_f16_batch28.txt)
    const long long tiles128 = ((M + 127) / 128) * ((c.CoutPad + 127) / 128);
    int t = (c.CoutPad >= 128 && tiles128 >= 192 && c.nchunks >= 32) ? TILE_PL128 : TILE_PL64;
    if (t == TILE_PL128 && mode == PREC_F16 && plh_ok(c)) t = TILE_PLH128;     // 3x3 / stride 1: the halo form beats the all-DMA tile wherever both run
    int s = 1;
    if (t == TILE_PL64) {
        const long long blocks = ((M + 63) / 64) * (c.CoutPad / 64);
        const int min_chunks = mode == PREC_F16 ? 8 : 4;
        while (blocks * s < 256 && c.nchunks / (s + 1) >= min_chunks && s < sk_max) ++s;
    }
    *tile = t; *splits = s;
}
#ifdef BP_EXPERIMENTAL   // BP_B3_PL64BD=1: bf16x3 on planes with the filters-direct plane tile (conv_pl.hip BDIR; A/B runs)
static void pl64_form(const ConvParams& c, int mode, int* tile) {
    static const bool bd = std::getenv("BP_B3_PL64BD") != nullptr;
    if (*tile == TILE_PL64 && mode == PREC_BF16X3 && c.wbd && bd) *tile = TILE_PL64BD;
}
#else
static void pl64_form(const ConvParams&, int, int*) {}
#endif

static void choose_h16(const ConvParams& c, long long M, int mode, int sk_max, int* tile, int* splits, bool lone = false) {
    // layers with operand planes (the fp16 mode; bf16x3 under BP_B3_PLANES) run on conv_pl.hip
#ifdef BP_EXPERIMENTAL   // BP_LEGACY=1: the fp32-activation kernels in every mode (A/B runs of the whole pipeline)
    static const bool legacy = std::getenv("BP_LEGACY") != nullptr;
#else
    constexpr bool legacy = false;
#endif
    if (conv_pl_eligible(c) && !legacy) { choose_pl(c, M, mode, sk_max, tile, splits); pl64_form(c, mode, tile); return; }
    if (mode == PREC_BF16X3)
        for (const PlanEntry& e : plan_file_entries())
            if (e.M == (int)M && e.CoutPad == c.CoutPad && e.nchunks == c.nchunks && !conv_tile_is_pl(e.tile) &&
                (!conv_tile_is_halo(e.tile) || conv_halo_eligible(c, e.tile))) { *tile = e.tile; *splits = e.splits; return; }
    static const bool halo_off = std::getenv("BP_NO_HALO") != nullptr;   // A/B runs: the round-3 plan (filters-direct kernel everywhere)
#ifdef BP_EXPERIMENTAL
    for (const PlanEntry* e = (mode == PREC_F16 ? kPlanF16 : kPlanB3); e->M != 0; ++e)   // tables end with a zero row
#else
    for (const PlanEntry* e = kPlanB3; e->M != 0; ++e)
#endif
        if (e->M == (int)M && e->CoutPad == c.CoutPad && e->nchunks == c.nchunks &&
            (!conv_tile_is_halo(e->tile) || (!halo_off && mode == PREC_BF16X3 && conv_halo_eligible(c, e->tile)))) {
            *tile = e->tile; *splits = e->splits;
            if (lone && e->tile == TILE_BD_K2)
                for (const PlanEntry* l = kPlanB3Lone; l->M != 0; ++l)
                    if (l->M == e->M && l->CoutPad == e->CoutPad && l->nchunks == e->nchunks) { *tile = l->tile; *splits = l->splits; break; }
            return;
        }
    int t = TILE_64x64_BD;   // bf16x3: the filters-direct 64x64 kernel at every batch size (profiles/r02_tune_b3_batch28.txt)
    // ... except the 3x3 / stride-1 layers of batched runs once one slice of halo tiles fills the chip: the 64x128 halo tile is
    // 1.2-1.5x the filters-direct kernel there (batch 28: 52x52 128 -> 256 217.6 against 319.0 us, 40x32 256 -> 512 378.8 against
    // 545.1 us; tools/bench_halo.py --batch 28, profiles/r04_halo_kernels.txt); BP_HALO_BATCH_MIN_TILES moves the threshold (A/B runs)
    static const int halo_min_tiles = std::getenv("BP_HALO_BATCH_MIN_TILES") ? std::atoi(std::getenv("BP_HALO_BATCH_MIN_TILES")) : 64;   // (64: batch 2 x 4 streams 1 118 -> 1 209, 4 x 3 1 274 -> 1 384, 28 x 2 1 509 -> 1 754 frames/s; 256 and 16 lose 4-8 % of that at batch 2 / 4)
    if (mode == PREC_BF16X3 && !halo_off && c.in16 == nullptr) {
        const int ht = conv_halo_eligible(c, TILE_HALO128) ? TILE_HALO128 : (conv_halo_eligible(c, TILE_HALO64K2) ? TILE_HALO64K2 : -1);
        if (ht >= 0 && ((M + 63) / 64) * (c.CoutPad / conv_tile_bn(ht)) >= halo_min_tiles) { *tile = ht; *splits = 1; return; }
    }
#ifdef BP_EXPERIMENTAL
    const long long tiles128 = ((M + 127) / 128) * ((c.CoutPad + 127) / 128);
    if (!(mode == PREC_BF16X3 && c.w16s)) t = (c.CoutPad >= 128 && tiles128 >= 128) ? TILE_W64_2x2 : TILE_64x64;
#endif
    const int bm = conv_tile_bm(t), bn = conv_tile_bn(t);
    const long long blocks = ((M + bm - 1) / bm) * ((c.CoutPad + bn - 1) / bn);
    const int target = t == TILE_W64_2x2 ? 256 : (mode == PREC_F16 ? 128 : 512);
    const int min_chunks = mode == PREC_F16 ? 8 : 4;
    int s = 1;
    while (blocks * s < target && c.nchunks / (s + 1) >= min_chunks && s < sk_max) ++s;
    *tile = t; *splits = s;
}

// a forced kernel id (bp_*_set_policy, tests and sweeps) applies to the layers it can run and is ignored for the others:
// the fp32-MFMA tiles run any layer (they read the fp32 activations), the operand-plane tiles the layers with planes
static bool tile_runs(int tile, const ConvParams& c, long long M = 0) {
    // a layer planned on the operand planes (in16 + wpl) may have NO fp32 input: plan_planes() dropped the fp32 store of producers
    // whose readers all take the planes.  The kernels that read fp32 activations are therefore never forced onto such a layer
    // (round-3 advisor finding: a forced tile 0 / 1 in the fp16 mode read tensors nobody stored)
    const bool on_planes = c.in16 != nullptr && c.wpl != nullptr;
    if (tile == TILE_64x64 || tile == TILE_128x64) return !on_planes;
    if (tile == TILE_S1) return conv_s1_eligible(c, M);
    if (tile == TILE_P3) return conv_p3_eligible(c, M);
    if (conv_tile_is_pl(tile)) return c.mfma_mode != PREC_F32 && conv_pl_eligible(c) && (!conv_tile_is_plh(tile) || conv_plh_eligible(c));
    if (tile == TILE_64x64_BD || tile == TILE_BD_K2) return c.mfma_mode == PREC_BF16X3 && conv_h16_eligible(c) && c.w16s != nullptr && !on_planes;
    if (conv_tile_is_halo(tile)) return c.mfma_mode == PREC_BF16X3 && c.in16 == nullptr && conv_halo_eligible(c, tile);
#ifdef BP_EXPERIMENTAL
    if (tile >= 0 && tile <= TILE_LAST) return c.mfma_mode != PREC_F32 && conv_h16_eligible(c);
#endif
    return false;
}

// the 3x3 / stride-1 RGB stem runs as a direct convolution (conv_igemm.hip stem3x3_kernel) unless a tile is forced;
// BP_NO_STEM3=1: on the fp32 MFMA kernel as before (A/B runs)
static bool stem3_wanted(const ConvParams& c, int force_tile) {
    static const bool off = std::getenv("BP_NO_STEM3") != nullptr;
    return !off && (force_tile < 0 || force_tile == TILE_STEM3) && conv_stem3_eligible(c);
}

// the 7x7 / stride-2 RGB stem runs on the fp16 matrix pipe when the ENGINE is in an fp16 mode (the layer itself is not 16-bit eligible: 3 input
// channels); BP_NO_STEM7=1: on the fp32 MFMA kernel as before (A/B runs)
static bool stem7_wanted(const ConvParams& c, int force_tile) {
    static const bool off = std::getenv("BP_NO_STEM7") != nullptr;
    return !off && force_tile < 0 && c.net_prec == PREC_F16 && conv_stem7_eligible(c);
}

static void choose_launch(const Op& op, int batch, int force_tile, int sk_target, int sk_min_chunks, int sk_max,
                          int* tile, int* splits, int* cps, bool lone = false) {
    const int mode = op.conv.mfma_mode;
    const ConvParams& c = op.conv;
    const long long M = (long long)batch * c.OH * c.OW;
    int t = TILE_64x64;   // fp32 MFMA kernel: 128x64 measured slower on every layer of both networks (tools/bench_conv.py)
    int s = 1;
    if (mode != PREC_F32) {
        choose_h16(c, M, mode, sk_max, &t, &s, lone);
        if (t == TILE_S1 && op.pool_out) t = TILE_PL64;     // (a plan-file row: the SE pool rides in the 64-row epilogue of the plane tile only)
        // round 5: the 1x1 layers of the batched fp16 runs (one K slice on a conv_pl tile, no SE pool in the epilogue) on the persistent
        // streaming kernel (conv_s1.hip); BP_NO_S1=1: the plane tiles as before (A/B runs)
        static const bool s1_off = std::getenv("BP_NO_S1") != nullptr;
        if (!s1_off && mode == PREC_F16 && s == 1 && conv_tile_is_pl(t) && !op.pool_out && conv_s1_eligible(c, M)) t = TILE_S1;
        // round 6: the 3x3 / stride-1 layers of the batched fp16 runs that were planned on the halo plane tile, on the persistent kernel
        // (conv_p3.hip); BP_NO_P3=1: the halo plane tile as before (A/B runs)
        static const bool p3_off = std::getenv("BP_NO_P3") != nullptr;
        if (!p3_off && mode == PREC_F16 && s == 1 && t == TILE_PLH128 && !op.pool_out && conv_p3_eligible(c, M)) t = TILE_P3;
        // ... and its 1x1 form (128-channel groups as the "halo", four chunks as the "taps") for the 1x1 layers with K >= 512, whether they were
        // planned on the plane tile or on the streaming kernel (28 frames, f16r, one launch at a time, profiles/r06_bench_p1.txt: 20x16 1 024 -> 256
        // 15.1 / 16.4 -> 12.3 us, 13x13 1 024 -> 512 15.0 / 16.4 -> 12.5, 26x26 512 -> 256 17.2 / 16.5 -> 13.4, 40x32 512 -> 128 17.7 / 16.8 -> 15.5;
        // at K = 256 the streaming kernel keeps its layers: 256 -> 1 024 16.3 against 17.9, 52x52 256 -> 128 15.7 against 16.4)
        // BP_P3_K1 = 0: off; 2: every eligible 1x1 layer (sweeps)
        static const int p3_k1 = std::getenv("BP_P3_K1") ? std::atoi(std::getenv("BP_P3_K1")) : 1;
        if (!p3_off && p3_k1 && mode == PREC_F16 && s == 1 && c.ksize == 1 && conv_tile_is_pl(t) && !op.pool_out && conv_p3_eligible(c, M) &&
            (p3_k1 == 2 || c.Cin >= 512)) t = TILE_P3;
        if (force_tile >= 0 && !((force_tile == TILE_S1 || force_tile == TILE_P3) && op.pool_out) && tile_runs(force_tile, c, M)) t = force_tile;
        if (!(sk_target == 512 && sk_min_chunks == 4 && sk_max == 8)) {   // explicit policy (tests, sweeps)
            const long long blocks = ((M + conv_tile_bm(t) - 1) / conv_tile_bm(t)) *
                                     ((c.CoutPad + conv_tile_bn(t) - 1) / conv_tile_bn(t));
            s = 1;
            while (blocks * s < sk_target && c.nchunks / (s + 1) >= sk_min_chunks && s < sk_max) ++s;
        }
        if (t == TILE_S1 || t == TILE_P3) s = 1;     // (a persistent grid: no K slices)
    } else if (stem3_wanted(c, force_tile)) {
        t = TILE_STEM3;       // the RGB 3x3 stem: direct convolution, no K slices
    } else if (stem7_wanted(c, force_tile)) {
        t = TILE_STEM7;       // the 7x7 / stride-2 RGB stem in the fp16 modes: fp16 MFMA over im2col rows in LDS, no K slices
    } else {
        if ((force_tile == TILE_64x64 || force_tile == TILE_128x64) && tile_runs(force_tile, c)) t = force_tile;
        const int bm = conv_tile_bm(t);
        const long long blocks = ((M + bm - 1) / bm) * (c.CoutPad / 64);
        while (blocks * s < sk_target && c.nchunks / (s + 1) >= sk_min_chunks && s < sk_max) ++s;
        if (t == TILE_64x64 && sk_target == 512 && sk_min_chunks == 4 && sk_max == 8)   // default policy: measured table
            for (const SplitEntry& e : kSplitTable)
                if (e.M == (int)M && e.CoutPad == c.CoutPad && e.nchunks == c.nchunks) { s = e.splits; break; }
    }
    {   // experiment hook (tools only): BP_SPLIT_PCT scales the slice count, e.g. 50 halves it
        static const int pct = std::getenv("BP_SPLIT_PCT") ? std::atoi(std::getenv("BP_SPLIT_PCT")) : 100;
        if (pct != 100 && s > 1) s = std::max(1, (s * pct + 50) / 100);
    }
    int per = 0;
    conv_split_plan(c, t, s, &s, &per);
    *tile = t; *splits = s; *cps = per;
}

int Net::add_conv(const std::string& name, const Tensor& in, const Tensor& out_view, const ConvWeights& cw, int Cout,
                  int k, int stride, int pad, int act, int store_mode, const Tensor* res, const float* res_scale,
                  int res_after_act, float bn_eps, int OH, int OW) {
    const int Cin = in.C;
    // K order: (ky, kx, ci) with `cpk` channels per tap -- Cin, or 4 for the RGB stems (Cin <= 4): a tap is then one
    // 16-B load (conv_igemm.hip, VEC mode 2) and the padded channel meets zero filter entries
    const int cpk = Cin <= 4 ? 4 : Cin;
    const int K = k * k * cpk;
    const int Kpad = round_up(K, 32);
    const int CoutPad = round_up(Cout, 64);
    float *dW, *dB;
    if (reuse_) {
        dW = upload_weights(nullptr, 0);
        dB = upload_weights(nullptr, 0);
    } else {
        std::vector<float> W((size_t)CoutPad * Kpad, 0.f), B(CoutPad, 0.f);
        for (int co = 0; co < Cout; ++co) {
            double s = 1.0, b = 0.0;
            if (cw.bn_scale) {
                s = darknet_bn_ ? (double)cw.bn_scale[co] / (std::sqrt((double)cw.bn_var[co]) + 1e-6)
                                : (double)cw.bn_scale[co] / std::sqrt((double)cw.bn_var[co] + (double)bn_eps);
                b = (double)cw.bn_bias[co] - (double)cw.bn_mean[co] * s;
            } else if (cw.bias) {
                b = cw.bias[co];
            }
            int n = co;
            if (store_mode == ST_PIXSHUF) n = (co & 3) * (Cout / 4) + (co >> 2);
            B[n] = (float)b;
            float* dst = W.data() + (size_t)n * Kpad;
            const float* src = cw.w + (size_t)co * Cin * k * k;
            for (int ci = 0; ci < Cin; ++ci)
                for (int t = 0; t < k * k; ++t) dst[(size_t)t * cpk + ci] = (float)((double)src[(size_t)ci * k * k + t] * s);
        }
        dW = upload_weights(W.data(), W.size());
        dB = upload_weights(B.data(), B.size());
    }

    Op op;
    op.type = OP_CONV;
    op.name = name;
    ConvParams& c = op.conv;
    c.in = in.p; c.in_ld = in.ld; c.N = 1; c.H = in.H; c.W = in.W; c.Cin = Cin;
    c.w = dW; c.Kpad = Kpad; c.Ktrue = K; c.bias = dB; c.cin_pack = cpk;
    c.out = out_view.p; c.out_ld = out_view.ld; c.OH = OH; c.OW = OW; c.Cout = Cout;
    c.ksize = k; c.stride = stride; c.pad = pad; c.act = act;
    c.res = res ? res->p : nullptr; c.res_ld = res ? res->ld : 0; c.res_scale = res_scale;
    c.res_after_act = res_after_act; c.store_mode = store_mode;
    c.M = OH * OW; c.nchunks = Kpad / 32; c.splits = 1; c.chunks_per_split = c.nchunks; c.partial = nullptr;
    c.tickets = nullptr;
    c.stamps = nullptr;
    c.w16 = nullptr; c.w16s = nullptr;
    c.in16 = nullptr; c.out16 = nullptr; c.wpl = nullptr; c.wbd = nullptr; c.in16_plane = c.out16_plane = 0; c.out_np = 0; c.abl = 0; c.skip_f32 = 0; c.res16 = nullptr; c.net_prec = PREC_F32;
    c.pool_out = nullptr; c.hy_full = c.hy_splits = c.hy_cps = 0;
    c.pf_ptr = nullptr; c.xcd_home = 0; c.xcc_of = nullptr; c.err_word = nullptr; c.tickets_local = nullptr; c.mtiles = c.n_tiles = c.work_blocks = c.pf_first = 0;
    c.pf_ntn = c.pf_splits = c.pf_cps = c.pf_nchunks = c.pf_tile_stride = c.pf_chunk_bytes = c.pf_cap = 0;
    c.CoutPad = CoutPad;
    const double Kalg = (double)k * k * Cin;   // algorithmic K (the packing pad is not work)
    op.flops = 2.0 * OH * OW * (double)Cout * Kalg;
    op.bytes = 4.0 * ((double)Cout * Kalg + (double)in.H * in.W * Cin + (double)OH * OW * Cout + (res ? (double)OH * OW * Cout : 0.0));
    ops_.push_back(op);
    return (int)ops_.size() - 1;
}

// split-K workspace for the launches the CURRENT plan (precision, tiles, policy) makes, over every batch size
size_t Net::workspace_need() const {
    size_t need = 0;
    for (const Op& op : ops_) {
        if (op.type != OP_CONV) continue;
        for (int lone = 0; lone < 2; ++lone)      // every batch size under the throughput plan, then under the lone-frame plan
            for (int b = 1; b <= max_batch_; ++b) {
                int tile, splits, cps;
                choose_launch(op, b, force_tile_, sk_target_, sk_min_chunks_, 64, &tile, &splits, &cps, lone != 0);
                // worst case over policies that may be set later: allow up to 64 splits at batch 1
                ConvParams q = op.conv; q.N = b; q.M = b * q.OH * q.OW; q.splits = splits;
                if (splits > 1) {
                    need = std::max(need, (size_t)splits * conv_tiles(q, tile) * conv_tile_bm(tile) * conv_tile_bn(tile));
                } else {   // a hybrid grid parks the slices of its last tiles (ConvParams::hy_*)
                    int full = 0, hs = 0, hcps = 0;
                    if (conv_hybrid_plan(q, tile, (size_t)-1, &full, &hs, &hcps))
                        need = std::max(need, (size_t)hs * (conv_tiles(q, tile) - full) * conv_tile_bm(tile) * conv_tile_bn(tile));
                }
            }
    }
    return std::max(need, (size_t)4 << 20);   // headroom so a later policy change can still split small layers
}

// ---- conv -> conv fusion (conv_fused.hip).  Structure first: op j is a 3x3 / stride-1 convolution whose input tensor is produced by a
// 1x1 convolution i and read by nobody else; optionally its own output is read only by a 1x1 convolution k.  (yolo/darknet.py:319-363:
// the 1x1 / 3x3 pair of a Darknet-53 residual block; SE_Resnet.py:25-42: conv1 / conv2 / conv3 of a bottleneck.)
void Net::find_fuse_groups() {
    fuse_groups_.clear();
    auto readers = [&](const float* t) {
        int n = 0;
        for (const Op& o : ops_) {
            if (o.type == OP_CONV) n += (o.conv.in == t) + (o.conv.res == t);
            else n += (o.a == t) + (o.b == t);
        }
        return n;
    };
    auto producer = [&](const float* t, int ld) {
        int idx = -1, n = 0;
        for (int i = 0; i < (int)ops_.size(); ++i) {
            const Op& o = ops_[i];
            if (o.type == OP_CONV ? (o.conv.out == t && o.conv.out_ld == ld) : o.out == t) { idx = i; ++n; }
        }
        return n == 1 ? idx : -1;
    };
    for (int j = 0; j < (int)ops_.size(); ++j) {
        const Op& c3 = ops_[j];
        if (c3.type != OP_CONV || c3.conv.ksize != 3 || c3.conv.stride != 1 || c3.conv.pad != 1) continue;
        const int i = producer(c3.conv.in, c3.conv.in_ld);
        if (i < 0 || i >= j || ops_[i].type != OP_CONV) continue;
        const ConvParams& pre = ops_[i].conv;
        if (!(pre.ksize == 1 && pre.stride == 1 && pre.res == nullptr && pre.res_scale == nullptr && pre.store_mode == ST_NHWC && pre.Cout == c3.conv.Cin)) continue;
        if (readers(pre.out) != 1 || ops_[i].pool_out) continue;
        if (i != j - 1) continue;                // (the members are consecutive launches in both networks: nothing runs between them)
        FuseGroup g{i, j, -1};
        if (j + 1 < (int)ops_.size() && ops_[j + 1].type == OP_CONV) {
            const ConvParams& post = ops_[j + 1].conv;
            if (post.in == c3.conv.out && post.in_ld == c3.conv.out_ld && post.ksize == 1 && post.stride == 1 && c3.conv.res == nullptr &&
                c3.conv.store_mode == ST_NHWC && readers(c3.conv.out) == 1 && !ops_[j].pool_out)
                g.post = j + 1;
        }
        fuse_groups_.push_back(g);
    }
}

// Which groups run fused under the current plan at this batch size: bf16x3 on the fp32-activation path, every member eligible for the
// fused kernel, enough patches to be worth a launch of their own.  A three-member group whose last 1x1 cannot join (the SE blocks'
// conv3 carries the average pool in its epilogue) falls back to its first two members.
void Net::plan_roles(int batch) {
    if (roles_batch_ == batch && roles_version_ == plan_version_) return;
    roles_.assign(ops_.size(), FR_NONE);
    role_group_.assign(ops_.size(), -1);
    roles_batch_ = batch; roles_version_ = plan_version_;
    static const bool env_off = std::getenv("BP_NO_FUSION") != nullptr;
    static const int min_blocks = env_int_early("BP_FUSE_MIN_BLOCKS", 48);
    if (!fusion_ || env_off || precision_ == PREC_F32 || force_tile_ >= 0) return;      // (bf16x3 on fp32 activations, fp16 on the operand planes)
    for (int gi = 0; gi < (int)fuse_groups_.size(); ++gi) {
        const FuseGroup& g = fuse_groups_[gi];
        ConvParams a, b, c;
        int ta, tb, tc;
        prepare_conv(ops_[g.pre], batch, a, ta);
        prepare_conv(ops_[g.c3], batch, b, tb);
        bool three = g.post >= 0;
        if (three) {
            prepare_conv(ops_[g.post], batch, c, tc);
            three = conv_fused_eligible(a, b, &c) && conv_fused_blocks(a, b, &c) >= min_blocks;
        }
        if (three) {
            roles_[g.pre] = roles_[g.c3] = FR_SKIP; roles_[g.post] = FR_HEAD3;
            role_group_[g.pre] = role_group_[g.c3] = role_group_[g.post] = gi;
        } else if (conv_fused_eligible(a, b, nullptr) && conv_fused_blocks(a, b, nullptr) >= min_blocks) {
            roles_[g.pre] = FR_SKIP; roles_[g.c3] = FR_HEAD2;
            role_group_[g.pre] = role_group_[g.c3] = gi;
        }
    }
}

int Net::fused_launches(int batch) {
    plan_roles(batch);
    int n = 0;
    for (int r : roles_) n += r == FR_HEAD2 || r == FR_HEAD3;
    return n;
}

void Net::finalize() {
    find_fuse_groups();
    (void)xcc_base();   // the one-time dispatch probe runs here (it allocates and copies: not inside a stream capture)
    partial_floats_ = workspace_need();
    partial_ = arena_.alloc(partial_floats_);
    // arrival counters for the in-kernel split-K reduction: one per output tile of the widest layer
    size_t tiles = 0;
    for (const Op& op : ops_)
        if (op.type == OP_CONV)
            tiles = std::max(tiles, (size_t)(((size_t)max_batch_ * op.conv.OH * op.conv.OW + 63) / 64) * (op.conv.CoutPad / 64));
    tickets_count_ = tiles;
    tickets_ = (int*)arena_.alloc_bytes(((2 + 64) * tiles + 1) * sizeof(int));      // [agent-scope counters | L2-local counters (xcd_home) | xcc_of | error word]
    BP_HIP(hipMemset(tickets_, 0, ((2 + 64) * tiles + 1) * sizeof(int)));
}

// The latency mode's XCD check (conv_dev.h xcd_home_verify): how many launches since the last call found a K slice on the wrong XCD
// (their tiles were NOT stored: the frame must be run again, without the mode).  Waits for `s`; clears the word.
int Net::take_xcd_errors(hipStream_t s) {
    if (!tickets_) return 0;
    int* w = tickets_ + (2 + 64) * tickets_count_;
    int v = 0;
    BP_HIP(hipMemcpyAsync(&v, w, sizeof(int), hipMemcpyDeviceToHost, s));
    BP_HIP(hipStreamSynchronize(s));
    if (v) {
        BP_HIP(hipMemsetAsync(w, 0, sizeof(int), s));
        BP_HIP(hipMemsetAsync(tickets_ + tickets_count_, 0, tickets_count_ * sizeof(int), s));   // (a skipped reducer re-armed its counter already; belt and braces)
        BP_HIP(hipStreamSynchronize(s));
    }
    return v;
}

void Net::set_precision(int prec) {
    BP_CHECK(prec == PREC_F32 || prec == PREC_F16 || prec == PREC_BF16X3 || prec == PREC_F16_RES, "unknown precision");
    f16_res_ = prec == PREC_F16_RES;
    if (f16_res_) prec = PREC_F16;
    // Two data paths for the 16-bit operand modes (A/B of the whole pipeline on one box, profiles/r03_ab_pipeline.txt):
    //   fp16   -> operand planes written by the producers, both operands by LDS-DMA (conv_pl.hip): +24 % at batch 1, +42 % at
    //             batch 28 over the round-2 kernels;
    //   bf16x3 -> fp32 activations, split in the consumer's K loop, filter fragments straight into registers (conv_igemm.hip,
    //             the filters-direct kernel): the plane path moves 6 bytes per activation element where this one moves 4 and
    //             holds 72 KB of LDS per block against 24.5 KB, and runs the pipeline 5 % SLOWER at batch 1 and at batch 28;
    //             BP_B3_PLANES=1 puts this mode on the plane path too (tests, A/B runs).
    for (Op& op : ops_)
        if (op.type == OP_CONV) { op.conv.w16 = nullptr; op.conv.w16s = nullptr; op.conv.net_prec = prec; }
    static const bool b3_planes = std::getenv("BP_B3_PLANES") != nullptr;
    const bool plane_path = prec == PREC_F16 || (prec == PREC_BF16X3 && b3_planes);
    // BP_B3_MIX=<pixels>: the per-layer form of that switch (round-5 verdict item 6) -- the 3x3 / stride-1 layers whose map has at most that many
    // pixels per image (the 22 layer-3 bottlenecks of the key-point detector: 20x16) read bf16x3 planes their 1x1 producers write, so that an
    // activation is split once instead of once per tap and N tile; every other layer stays on the fp32-activation kernels
    static const int b3_mix = std::getenv("BP_B3_MIX") ? std::atoi(std::getenv("BP_B3_MIX")) : 0;
    const bool mixed = prec == PREC_BF16X3 && !plane_path && b3_mix > 0;
    if (prec != PREC_F32) {
        std::lock_guard<std::mutex> lk(store_->f16_mutex);
        bool made = false;
        for (Op& op : ops_) {
            if (op.type != OP_CONV) continue;
            ConvParams& c = op.conv;
            if (!((c.Cin % 32 == 0) && (c.in_ld % 4 == 0) && c.ksize <= 8)) continue;   // the RGB stems stay on the fp32 kernel
            const size_t planes = prec == PREC_F16 ? 1 : 3;
#ifdef BP_EXPERIMENTAL   // unstaged copies: the LDS-staged round-1/2 kernels and conv_w64.hip (experimental library only)
            {
                auto& copies = prec == PREC_F16 ? store_->f16 : store_->bf16x3;
                auto it = copies.find(c.w);
                if (it == copies.end()) {
                    const size_t n = (size_t)c.CoutPad * c.Kpad;
                    unsigned short* d = (unsigned short*)store_->arena.alloc_bytes(planes * n * sizeof(unsigned short));
                    if (prec == PREC_F16) launch_f32_to_f16(c.w, d, (long long)n, nullptr);
                    else launch_f32_to_bf16x3(c.w, d, (long long)n, nullptr);
                    it = copies.emplace(c.w, d).first;
                    made = true;
                }
                c.w16 = it->second;
            }
            const bool want_staged = true;
#else
            const bool want_staged = prec == PREC_BF16X3 && !plane_path;
#endif
            if (want_staged) {   // stage-packed copy: the filters-direct kernel's fragment image
                auto& staged = prec == PREC_F16 ? store_->f16s : store_->bf16x3s;
                auto is = staged.find(c.w);
                if (is == staged.end()) {
                    unsigned short* d = (unsigned short*)store_->arena.alloc_bytes(planes * c.CoutPad * c.Kpad * sizeof(unsigned short));
#ifdef BP_EXPERIMENTAL
                    if (prec == PREC_F16) launch_f32_to_f16_staged(c.w, d, c.CoutPad, c.Kpad, nullptr);
                    else
#endif
                    launch_f32_to_bf16x3_staged(c.w, d, c.CoutPad, c.Kpad, nullptr);
                    is = staged.emplace(c.w, d).first;
                    made = true;
                }
                c.w16s = is->second;
            }
        }
        if (made) {
            BP_HIP(hipGetLastError());
            BP_HIP(hipDeviceSynchronize());
        }
    }
#ifdef BP_EXPERIMENTAL   // timing experiment (wrong results): every layer reads ONE filter buffer, so the filters are always cache-resident
    if (std::getenv("BP_ALIAS_WEIGHTS")) {
        const unsigned short* big = nullptr; size_t big_n = 0;
        for (Op& op : ops_)
            if (op.type == OP_CONV && op.conv.w16s && (size_t)op.conv.CoutPad * op.conv.Kpad > big_n) { big_n = (size_t)op.conv.CoutPad * op.conv.Kpad; big = op.conv.w16s; }
        for (Op& op : ops_)
            if (op.type == OP_CONV && op.conv.w16s) op.conv.w16s = big;
    }
#endif
    plan_planes(plane_path || mixed ? prec : PREC_F32, mixed ? b3_mix : 0);
    for (Op& op : ops_)
        if (op.type == OP_CONV) op.conv.mfma_mode = (prec != PREC_F32 && (conv_h16_eligible(op.conv) || conv_pl_eligible(op.conv))) ? prec : PREC_F32;
    precision_ = prec;
    ++plan_version_;
    // the 16-bit plans split differently from the fp32 one the workspace was first sized for
    if (const size_t need = workspace_need(); need > partial_floats_) {
        partial_floats_ = need;
        partial_ = arena_.alloc(need);
    }
}

Net::ActAlloc* Net::find_act(const float* p) {
    for (ActAlloc& a : acts_)
        if (p >= a.base && p < a.base + a.elems) return &a;
    return nullptr;
}

// Operand planes (conv_pl.hip): every activation allocation some 16-bit-eligible convolution reads gets three bf16 planes
// (the fp16 mode uses the first), every producer of such an allocation -- convolutions of any arithmetic through
// conv_tail.inc, the pooling / shuffle / copy kernels through a conversion launch -- is pointed at them, and the
// filters are packed into the kernel's LDS image.  Planes are allocated once per engine (first 16-bit mode), outside any
// graph capture.
void Net::plan_planes(int prec, int mix_hw) {
    for (Op& op : ops_) {
        op.out16 = nullptr; op.out16_plane = 0;
        if (op.type == OP_CONV) { op.conv.in16 = nullptr; op.conv.out16 = nullptr; op.conv.wpl = nullptr; op.conv.wbd = nullptr; op.conv.out_np = 0; op.conv.in16_plane = op.conv.out16_plane = 0; op.conv.skip_f32 = 0; op.conv.res16 = nullptr; }
    }
    for (ActAlloc& a : acts_) { a.f32_read = true; a.wanted = false; }
    if (prec == PREC_F32) return;
#ifdef BP_EXPERIMENTAL
    if (std::getenv("BP_LEGACY")) return;   // A/B runs: the fp32-activation data path in every mode
#endif
    const int np = prec == PREC_F16 ? 1 : 3;
    auto pl_shape_ok = [mix_hw](const ConvParams& c) {
        if (mix_hw > 0 && !(c.ksize == 3 && c.stride == 1 && c.OH * c.OW <= mix_hw)) return false;   // (BP_B3_MIX: only the small 3x3 layers)
        return (c.Cin % 32 == 0) && (c.in_ld % 8 == 0) && c.ksize * c.ksize <= 32;
    };
    bool made = false;
    for (Op& op : ops_) {
        if (op.type != OP_CONV || !pl_shape_ok(op.conv)) continue;
        ActAlloc* a = find_act(op.conv.in);
        if (!a || (op.conv.in - a->base) % 8 != 0 || a->elems % 8 != 0) continue;
        a->wanted = true;
    }
    // fp16 skip connections (PREC_F16_RES): a tensor that is ONLY ever a skip connection -- the SE blocks' T = bn3(conv3), which the
    // "downsample" convolution adds scaled (SE_Resnet.py:31-40) -- gets an fp16 plane too, so that it travels as 2 bytes per element like every
    // other skip connection of the mode (round 5: at 28 frames per launch layer1.0's T was 147 MB written and 147 MB read as fp32)
    if (f16_res_ && np == 1)
        for (Op& op : ops_)
            if (op.type == OP_CONV && op.conv.res)
                if (ActAlloc* r = find_act(op.conv.res); r && (op.conv.res - r->base) % 8 == 0 && r->elems % 8 == 0 && (op.conv.res_ld & 7) == 0) r->wanted = true;
    for (ActAlloc& a : acts_)
        if (a.wanted && !a.planes) {
            a.planes = (unsigned short*)arena_.alloc_bytes(3 * a.elems * sizeof(unsigned short));
            BP_HIP(hipMemset(a.planes, 0, 3 * a.elems * sizeof(unsigned short)));
        }
    std::lock_guard<std::mutex> lk(store_->f16_mutex);
    auto& packed = np == 1 ? store_->wpl1 : store_->wpl3;
    for (Op& op : ops_) {
        if (op.type == OP_CONV) {
            ConvParams& c = op.conv;
            if (ActAlloc* o = find_act(c.out); o && o->planes && o->wanted) {
                c.out16 = o->planes + (c.out - o->base);
                c.out16_plane = (long long)o->elems;
                c.out_np = np;
            }
            if (!pl_shape_ok(c)) continue;
            ActAlloc* a = find_act(c.in);
            if (!a || !a->planes || !a->wanted || (c.in - a->base) % 8 != 0) continue;
            c.in16 = a->planes + (c.in - a->base);
            c.in16_plane = (long long)a->elems;
            auto it = packed.find(c.w);
            if (it == packed.end()) {
                unsigned short* d = (unsigned short*)store_->arena.alloc_bytes((size_t)np * c.CoutPad * c.Kpad * sizeof(unsigned short));
                launch_pack_wpl(c.w, d, c.CoutPad, c.Kpad, c.Cin, c.ksize, np, nullptr);
                it = packed.emplace(c.w, d).first;
                made = true;
            }
            c.wpl = it->second;
#ifdef BP_EXPERIMENTAL
            if (np == 3 && std::getenv("BP_B3_PL64BD")) {   // the filters-direct plane tile's fragment image (TILE_PL64BD)
                auto ib = store_->wbd3.find(c.w);
                if (ib == store_->wbd3.end()) {
                    unsigned short* d = (unsigned short*)store_->arena.alloc_bytes((size_t)3 * c.CoutPad * c.Kpad * sizeof(unsigned short));
                    launch_f32_to_bf16x3_staged(c.w, d, c.CoutPad, c.Kpad, nullptr, c.Cin);
                    ib = store_->wbd3.emplace(c.w, d).first;
                    made = true;
                }
                c.wbd = ib->second;
            }
#endif
        } else if (op.out) {
            if (ActAlloc* o = find_act(op.out); o && o->planes && o->wanted) {
                op.out16 = o->planes + (op.out - o->base);
                op.out16_plane = (long long)o->elems;
            }
        }
    }
    if (made) {
        BP_HIP(hipGetLastError());
        BP_HIP(hipDeviceSynchronize());
    }
    // Which fp32 tensors does anything still read?  Residual operands, the inputs of the pooling / shuffle / add kernels
    // and of convolutions that run on the fp32 kernels; everything else that has planes is read through them only, and
    // its producers drop the fp32 store (4 of 10 bytes per element in the bf16x3 mode, 4 of 6 in the fp16 mode; the
    // dirty lines a kernel leaves behind are written back before its successor starts).  BP_KEEP_F32=1 keeps them all.
    static const bool keep_all = std::getenv("BP_KEEP_F32") != nullptr;
    for (ActAlloc& a : acts_) a.f32_read = keep_all || a.planes == nullptr || !a.wanted;
    auto mark = [&](const float* q) { if (q) if (ActAlloc* a = find_act(q)) a->f32_read = true; };
    // fp16 skip connections (PREC_F16_RES): a residual whose tensor has an fp16 plane -- some convolution reads it as input, so its
    // producer writes the plane anyway -- is read from that plane (2 B per element instead of 4) and does not keep the fp32 tensor alive
    if (f16_res_ && np == 1)
        for (Op& op : ops_)
            if (op.type == OP_CONV && op.conv.res)
                if (ActAlloc* r = find_act(op.conv.res); r && r->planes && (op.conv.res - r->base) % 4 == 0 && (op.conv.res_ld & 3) == 0)
                    op.conv.res16 = r->planes + (op.conv.res - r->base);
    for (const Op& op : ops_) {
        if (op.type == OP_CONV) {
            if (!op.conv.res16) mark(op.conv.res);
            if (!(op.conv.in16 && op.conv.wpl)) mark(op.conv.in);
        } else if (op.type == OP_AVGPOOL && &op != ops_.data() && (&op - 1)->type == OP_CONV && (&op - 1)->pool_out == op.out) {
            // the SE average pool normally rides in the producing convolution's epilogue (pool_in_epilogue: fp32 sums of the accumulators,
            // nothing reads the tensor); where it cannot (a tile that spans images), run_op rebuilds the fp32 tensor from the plane first
        } else {
            mark(op.a); mark(op.b);
        }
    }
    for (Op& op : ops_)
        if (op.type == OP_CONV && op.conv.out16)
            if (ActAlloc* o = find_act(op.conv.out)) op.conv.skip_f32 = o->f32_read ? 0 : 1;
}

static int planes_np(int prec) { return prec == PREC_F16 ? 1 : 3; }

// filter prefetch (ConvParams::pf_*): one extra block per (N-tile, K-slice) pair of the next launch pulls at most this much
static int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }
static const int kPrefetchCap = env_int("BP_PF_CAP_KB", 128) * 1024;
// XCC_ID of block 0 of a launch, or -1 when the dispatch is not the round robin ConvParams::xcd_home relies on (or
// BP_NO_XCD_HOME=1): one probe launch per process, 64 blocks, every block b must report (base + b) % 8
int xcc_base() {
    // one probe per DEVICE (round-3 advisor finding: a process that drives several GPUs must not reuse the first device's answer)
    static std::mutex mu;
    static std::map<int, int> by_device;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(mu);
    auto it = by_device.find(dev);
    if (it != by_device.end()) return it->second;
    const int base = [] {
        if (std::getenv("BP_NO_XCD_HOME")) return -1;
        const int blocks = 64;
        int* d = nullptr;
        if (hipMalloc(&d, 2 * blocks * sizeof(int)) != hipSuccess) return -1;
        launch_probe_placement(d, blocks, nullptr);
        std::vector<int> h(2 * blocks);
        const bool ok = hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        if (!ok) return -1;
        for (int b = 0; b < blocks; ++b)
            if (((h[2 * b] - h[0] - b) & 7) != 0) return -1;
        return h[0] & 7;
    }();
    by_device[dev] = base;
    return base;
}

// p's launch carries the prefetch blocks for `next` (launched with tile nt, ns K slices of nc chunks), when next's work
// blocks of residue x read the N-tiles n == x (mod min(N-tiles, 8)) -- the xcd_home layout, or the plain one-slice grid of the
// 64x64 filters-direct kernel -- from a filter image that is contiguous per (N-tile, K-slice) pair
// Hybrid grid of a one-slice conv_pl launch (ConvParams::hy_*): when the launch is ONE block per CU plus a few more (256 < tiles
// <= 422 on 256 CUs), 256 tiles run whole and the rest are cut along K so that they spread over every CU instead of doubling
// up on a few.  Measured at batch 28, fp16 (profiles/r03_hybrid_grid.txt): 296 tiles of 256x128 72.2 -> 65.9 us, 280 tiles
// 124.8 -> 106.1 us, 296 tiles of 128x128 72.5 -> 67.5 us; with several blocks per CU in flight the dispatcher balances the
// tail by itself and the cut only adds its reduction (1 184 tiles of 128x128: 79.8 -> 98.8 us), so longer grids stay whole.
// In the PIPELINE it loses -- fp16 batch 28 x 3 streams 3 940-4 010 against 4 060-4 140 frames/s, the other runs unchanged: with
// other streams' blocks on the CUs there is no "one block per CU" to complete -- so it is OFF unless BP_HYBRID=1 (A/B runs, tests).
bool conv_hybrid_plan(const ConvParams& p, int tile, size_t partial_floats, int* full, int* hs, int* hcps) {
    // (read per call on purpose: tests/test_gpu_conv.py::test_conv_pl_hybrid_grid toggles it inside one process; the lookup only runs
    // for one-slice conv_pl launches in eager mode and at graph capture, never in a graph replay)
    const bool off = std::getenv("BP_HYBRID") == nullptr;
    if (off || !conv_tile_is_pl(tile) || p.splits != 1 || p.xcd_home || p.nchunks < 16) return false;
    if (!(tile == TILE_PL64 || tile == TILE_PL128 || tile == TILE_PL128x64 || tile == TILE_PL256x128)) return false;
    const int T = conv_tiles(p, tile), unit = 256;
    const int rem = T - unit;
    if (rem <= 0 || rem * 100 > unit * 65) return false;
    int s = std::min(std::min(unit / rem, 8), p.nchunks / 8);
    if (s < 2) return false;
    const int cps = (p.nchunks + s - 1) / s;
    s = (p.nchunks + cps - 1) / cps;
    if ((size_t)s * rem * conv_tile_bm(tile) * conv_tile_bn(tile) > partial_floats) return false;
    *full = unit; *hs = s; *hcps = cps;
    return true;
}

bool conv_home_layout(int tile, int splits) {
    return splits > 1 && splits <= 64 && xcc_base() >= 0 && (tile == TILE_64x64_BD || (conv_tile_is_pl(tile) && !conv_tile_is_plh(tile)));
}
void conv_prefetch_of(ConvParams& p, const ConvParams& next, int nt, int ns, int nc) {
    p.pf_ptr = nullptr;
    const bool plbd = nt == TILE_PL64BD && next.mfma_mode == PREC_BF16X3 && next.wbd && conv_home_layout(nt, ns);
    const bool pl = plbd || (next.mfma_mode != PREC_F32 && (nt == TILE_PL64) && next.wpl && conv_home_layout(nt, ns));
    const bool bd = nt == TILE_64x64_BD && next.mfma_mode == PREC_BF16X3 && next.w16s && (ns == 1 || conv_home_layout(nt, ns));
    const int ntn = (next.CoutPad + 63) / 64;
    if (!(pl || bd) || (ntn & (ntn - 1)) != 0 || xcc_base() < 0) return;
    const int np = planes_np(next.mfma_mode);
    p.pf_ptr = plbd ? (const void*)next.wbd : pl ? (const void*)next.wpl : (const void*)next.w16s;
    p.pf_ntn = ntn; p.pf_splits = ns; p.pf_cps = nc; p.pf_nchunks = next.nchunks;
    p.pf_chunk_bytes = np * 4096;                  // 64 filter rows x 32 k x 2 B per plane
    p.pf_tile_stride = next.nchunks * p.pf_chunk_bytes;
    p.pf_cap = kPrefetchCap;
}

// The SE blocks' average pool inside the producing conv's epilogue (ConvParams::pool_out): for the 64-row tiles with the
// staged vector epilogue, when a tile never spans two images.  BP_NO_POOL_FUSION=1: the separate kernel (A/B, tests).
bool Net::pool_in_epilogue(const Op& conv, int batch, int tile) const {
    static const bool off = std::getenv("BP_NO_POOL_FUSION") != nullptr;
    const ConvParams& c = conv.conv;
    if (off || !conv.pool_out || conv_tile_bm(tile) != 64 || conv_tile_bn(tile) != 64) return false;
    if (c.store_mode != ST_NHWC || c.res || c.res_scale || c.act != ACT_LINEAR || (c.out_ld & 3) || (c.Cout & 3)) return false;
    return batch == 1 || (c.OH * c.OW) % 64 == 0;
}
bool Net::pooled_by_conv(const Op& pool, int batch) const {
    if (&pool == ops_.data()) return false;
    const Op& prev = *(&pool - 1);
    if (prev.type != OP_CONV || prev.pool_out != pool.out) return false;
    int tile, splits, cps;
    choose_launch(prev, batch, force_tile_, sk_target_, sk_min_chunks_, sk_max_splits_, &tile, &splits, &cps, prefetch_);
    return pool_in_epilogue(prev, batch, tile);
}

// the launch descriptor of convolution `op` at this batch size under the current plan: tile, K slices, workspaces, epilogue extras
void Net::prepare_conv(const Op& op, int batch, ConvParams& p, int& tile) {
    {
        {
            p = op.conv;
            p.N = batch;
            p.M = batch * p.OH * p.OW;
            int splits, cps;
            choose_launch(op, batch, force_tile_, sk_target_, sk_min_chunks_, sk_max_splits_, &tile, &splits, &cps, prefetch_);
            const int planned = splits;
            while (splits > 1 && (size_t)splits * conv_tiles(p, tile) * conv_tile_bm(tile) * conv_tile_bn(tile) > partial_floats_) {
                conv_split_plan(p, tile, splits - 1, &splits, &cps);
            }
            if (splits != planned) {   // (a policy / plan file set after finalize() may ask for more slab space than the workspace holds: say so once)
                static bool warned = false;
                if (!warned) {
                    warned = true;
                    std::fprintf(stderr, "betapose_hip: %s planned with %d K slices runs with %d (split-K workspace of %zu floats)\n",
                                 op.name.c_str(), planned, splits, partial_floats_);
                }
            }
            p.splits = splits; p.chunks_per_split = cps; p.partial = partial_; p.tickets = tickets_;
            p.tickets_local = tickets_ + tickets_count_;
            p.xcc_of = tickets_ + 2 * tickets_count_;
            p.err_word = tickets_ + (2 + 64) * tickets_count_;
            p.stamps = nullptr;
            if (stamps_) {   // in-situ timing (set_stamps): this conv's region of the stamp buffer, when its grid fits
                int ord = 0;
                for (const Op* q = ops_.data(); q != &op; ++q) ord += q->type == OP_CONV;
                if ((long long)conv_tiles(p, tile) * splits <= stamp_slots_) p.stamps = stamps_ + ((size_t)ord * stamp_slots_) * 8;
            }
            // lone-frame latency mode (set_prefetch; bp_common.h "launch layout by XCD"): all K slices of a tile on one XCD with
            // the hand-off through its L2 (xcd_home), and blocks that pull the next convolution's filters into the L2 that
            // will read them (pf_*; a hint: a wrong guess about the next launch costs bandwidth, not correctness).  One frame
            // at a time: 377 -> 384 -> 395 frames/s (fp16 533 -> 545 -> 558); with four in flight 916 -> 908 -> 895, so it is
            // a mode, not the default (profiles/r03_prefetch_ab.txt)
            p.xcd_home = (prefetch_ && conv_home_layout(tile, splits)) ? (std::getenv("BP_XCD_FAULT") ? 3 : 1) : 0;   // (3: the tests' fault injection, conv_dev.h xcd_home_mark)   // (with four frames in flight it gains nothing even on launches whose tiles divide evenly over the XCDs: 898 against 896)
            p.pool_out = pool_in_epilogue(op, batch, tile) ? op.pool_out : nullptr;
            p.hy_splits = 0;
            if (splits == 1 && !p.pool_out && conv_hybrid_plan(p, tile, partial_floats_, &p.hy_full, &p.hy_splits, &p.hy_cps)) {
                if ((long long)(conv_tiles(p, tile) - p.hy_full) > (long long)tickets_count_) p.hy_splits = 0;
            }
            p.pf_ptr = nullptr;
            if (prefetch_ && (tile == TILE_64x64_BD || conv_tile_is_pl(tile))) {
                for (const Op* q = &op + 1; q != ops_.data() + ops_.size(); ++q) {
                    if (q->type != OP_CONV) continue;
                    ConvParams n = q->conv;
                    n.N = batch; n.M = batch * n.OH * n.OW;
                    int nt, ns, nc;
                    choose_launch(*q, batch, force_tile_, sk_target_, sk_min_chunks_, sk_max_splits_, &nt, &ns, &nc, prefetch_);
                    conv_prefetch_of(p, n, nt, ns, nc);
                    break;
                }
            }
        }
    }
}

#ifdef BP_EXPERIMENTAL
// xcd mode (mega.inc): the convolution launches of one pass, in order, as descriptors for the persistent kernel
void Net::emit_conv_ops(int batch, std::vector<MegaOp>& out) {
    BP_CHECK(batch >= 1 && batch <= max_batch_, "batch out of range");
    for (const Op& op : ops_) {
        if (op.type != OP_CONV) continue;
        ConvParams p;
        int tile;
        prepare_conv(op, batch, p, tile);
        MegaOp o;
        mega_make_conv_op(p, tile, &o);
        out.push_back(o);
    }
}
#endif

void Net::run_op(const Op& op, int batch, hipStream_t s) {
    if (op.type == OP_CONV) {
        plan_roles(batch);
        const int idx = (int)(&op - ops_.data());
        const int role = roles_[idx];
        if (role == FR_SKIP) return;                       // computed inside the group's one launch (at its last member)
        if (role == FR_HEAD2 || role == FR_HEAD3) {
            const FuseGroup& g = fuse_groups_[role_group_[idx]];
            ConvParams a, b, c;
            int ta, tb, tc;
            prepare_conv(ops_[g.pre], batch, a, ta);
            prepare_conv(ops_[g.c3], batch, b, tb);
            if (role == FR_HEAD3) prepare_conv(ops_[g.post], batch, c, tc);
            // in-situ stamps: prepare_conv sized the last member's region by ITS OWN tile grid; the fused kernel stamps one slot per
            // patch block (stamps[blockIdx.x * 8 + k]), which can be far more (28 x 676 patches against 4 732 tiles) and would run
            // into the following convolutions' regions -- no marks for this launch unless its grid fits
            if (stamps_ && conv_fused_blocks(a, b, role == FR_HEAD3 ? &c : nullptr) > stamp_slots_) b.stamps = c.stamps = nullptr;
            launch_conv_fused(a, b, role == FR_HEAD3 ? &c : nullptr, s);
            return;
        }
    }
    run_op_unfused(op, batch, s);
}

void Net::run_op_unfused(const Op& op, int batch, hipStream_t s) {
    switch (op.type) {
        case OP_CONV: {
            ConvParams p;
            int tile;
            prepare_conv(op, batch, p, tile);
            launch_conv(p, tile, s);
        } break;
        // (producers that are not convolutions: the pooling / shuffle kernels write their operand planes themselves; the
        // unfused add / upsample / copy fall-backs, which the two networks' default cfgs never emit, convert behind them)
        case OP_MAXPOOL:
            launch_maxpool3s2p1(op.a, op.out, batch, op.H, op.W, op.C, op.OH, op.OW, s, op.out16, op.out16_plane, planes_np(precision_));
            break;
        case OP_ADD:
            launch_add(op.a, op.a_ld, op.b, op.b_ld, op.out, op.out_ld, (long long)batch * op.H * op.W, op.C, s);
            if (op.out16) launch_f32_to_planes(op.out, op.out_ld, (long long)batch * op.H * op.W, op.C, op.out16, op.out16_plane, planes_np(precision_), s);
            break;
        case OP_UPSAMPLE:
            launch_upsample2(op.a, op.a_ld, op.out, op.out_ld, batch, op.H, op.W, op.C, s);
            if (op.out16) launch_f32_to_planes(op.out, op.out_ld, (long long)batch * 4 * op.H * op.W, op.C, op.out16, op.out16_plane, planes_np(precision_), s);
            break;
        case OP_COPYCH:
            launch_copy_channels(op.a, op.a_ld, op.out, op.out_ld, (long long)batch * op.H * op.W, op.C, s);
            if (op.out16) launch_f32_to_planes(op.out, op.out_ld, (long long)batch * op.H * op.W, op.C, op.out16, op.out16_plane, planes_np(precision_), s);
            break;
        case OP_PIXSHUF:
            launch_pixel_shuffle2(op.a, op.out, batch, op.H, op.W, op.C, s, op.out16, op.out16_plane, planes_np(precision_));
            break;
        case OP_AVGPOOL:
            if (pooled_by_conv(op, batch)) break;      // its slice sums were written by the producing conv's epilogue
            // (a tensor whose fp32 store was dropped because only its plane is read -- plan_planes -- is rebuilt from the plane for this kernel)
            if (ActAlloc* a = find_act(op.a); a && a->planes && !a->f32_read && precision_ != PREC_F32)
                launch_planes_to_f32(a->planes + (op.a - a->base), (long long)a->elems, precision_ == PREC_F16 ? 1 : 3, const_cast<float*>(op.a), op.a_ld,
                                     (long long)batch * op.H * op.W, op.C, s);
            launch_avgpool(op.a, op.a_ld, op.out, batch, op.H * op.W, op.C, s);
            break;
        case OP_FC: {
            int parts = op.in_parts;
            if (&op != ops_.data() && (&op - 1)->type == OP_AVGPOOL && (&op - 1)->out == op.a && pooled_by_conv(*(&op - 1), batch))
                parts = ((&op - 1)->H * (&op - 1)->W + 63) / 64;
            launch_fc(op.a, op.w, op.bias, op.out, batch, op.Cin, op.Cout, op.act, parts, op.in_scale, s);
        } break;
        default:
            throw Error("unknown op");
    }
}

void Net::run_ops(int batch, hipStream_t s) {
    BP_CHECK(batch >= 1 && batch <= max_batch_, "batch out of range");
    for (const Op& op : ops_) run_op(op, batch, s);
    BP_HIP(hipGetLastError());
}

int Net::profile(int batch, int iters, float* ms, int* info, int cap, hipStream_t s) {
    const int n = (int)ops_.size();
    if (!ms || cap < n) return n;
    BP_CHECK(batch >= 1 && batch <= max_batch_ && iters >= 1, "profile arguments");
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) BP_HIP(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    run_ops(batch, s);   // warm
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < n; ++i) {
            plan_roles(batch);
            if (ops_[i].type == OP_CONV && roles_[i] != FR_SKIP) {      // (a member computed inside its block's one launch launches nothing: empty bracket)
                ConvProfHook hook{ev[2 * i], ev[2 * i + 1]};
                g_conv_prof = &hook;
                try { run_op(ops_[i], batch, s); } catch (...) { g_conv_prof = nullptr; throw; }
                g_conv_prof = nullptr;
            } else {
                BP_HIP(hipEventRecord(ev[2 * i], s));
                run_op(ops_[i], batch, s);
                BP_HIP(hipEventRecord(ev[2 * i + 1], s));
            }
        }
        BP_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) {
            float t = 0;
            BP_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
            acc[i] += t;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < n; ++i) {
        ms[i] = (float)(acc[i] / iters);
        // an average pool whose slice sums ride in the producing conv's epilogue launches nothing: 0 ms, info tile = -1
        const bool fused_away = ops_[i].type == OP_AVGPOOL && pooled_by_conv(ops_[i], batch);
        if (fused_away) ms[i] = 0.f;
        if (info) {
            int tile = 0, splits = 1, cps = 0, vec = 0, conv = ops_[i].type == OP_CONV;
            if (conv) {
                choose_launch(ops_[i], batch, force_tile_, sk_target_, sk_min_chunks_, sk_max_splits_, &tile, &splits, &cps, prefetch_);
                vec = conv_vec_mode(ops_[i].conv) ? 1 : 0;
                if (ops_[i].conv.mfma_mode != PREC_F32 && tile != TILE_64x64 && tile != TILE_128x64)
                    vec = 1 + ops_[i].conv.mfma_mode;   // 2 fp16 operands, 3 bf16x3 operands
            }
            // members of a fused block (conv_fused.hip): the skipped ones launch nothing (tile -1, like the pooled average pools), the
            // last one carries the block's one launch (TILE_FUSED); their FLOPs / bytes belong to that launch (bench.py adds them up)
            if (conv) {
                plan_roles(batch);
                if (roles_[i] == FR_SKIP) { tile = -1; ms[i] = 0.f; }
                else if (roles_[i] == FR_HEAD2 || roles_[i] == FR_HEAD3) { tile = TILE_FUSED; splits = 1; vec = 1 + ops_[i].conv.mfma_mode; }
            }
            info[4 * i] = conv; info[4 * i + 1] = fused_away ? -1 : tile; info[4 * i + 2] = vec; info[4 * i + 3] = splits;
        }
    }
    return n;
}

void Net::tap_copy(int i, int batch, float* d_out_nchw, hipStream_t s) {
    BP_CHECK(i >= 0 && i < (int)taps_.size(), "tap index");
    const Tensor& t = taps_[i];
    // a tensor that lives only inside a fused block (conv_fused.hip) is rebuilt here by the unfused launches of the members that make it:
    // their inputs are intact (one allocation per layer output) -- the tap then shows what the UNFUSED kernels compute from the same input
    plan_roles(batch);
    for (int j = 0; j < (int)ops_.size(); ++j)
        if (ops_[j].type == OP_CONV && roles_[j] == FR_SKIP && ops_[j].conv.out == t.p) {
            const FuseGroup& g = fuse_groups_[role_group_[j]];
            run_op_unfused(ops_[g.pre], batch, s);
            if (j == g.c3) run_op_unfused(ops_[g.c3], batch, s);
        }
    // a tensor whose producers dropped the fp32 store (plan_planes) is rebuilt from its planes first: exact in the
    // bf16x3 mode (the planes ARE the fp32 value), the fp16-rounded value in the fp16 mode
    if (ActAlloc* a = find_act(t.p); a && a->planes && !a->f32_read && precision_ != PREC_F32)
        launch_planes_to_f32(a->planes + (t.p - a->base), (long long)a->elems, precision_ == PREC_F16 ? 1 : 3, t.p, t.ld,
                             (long long)batch * t.H * t.W, t.C, s);
    launch_nhwc_to_nchw(t.p, t.ld, d_out_nchw, batch, t.C, t.H, t.W, s);
    BP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ cfg parsing (yolo/darknet.py:45-74 semantics)
struct CfgBlock {
    std::string type;
    std::map<std::string, std::string> kv;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    int geti(const std::string& k, int def) const {
        auto it = kv.find(k);
        return it == kv.end() ? def : std::atoi(it->second.c_str());
    }
    std::string gets(const std::string& k) const {
        auto it = kv.find(k);
        return it == kv.end() ? std::string() : it->second;
    }
};
static std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
static std::vector<CfgBlock> parse_cfg(const std::string& text) {
    std::vector<CfgBlock> out;
    std::istringstream is(text);
    std::string line;
    while (std::getline(is, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        if (line[0] == '[') {
            CfgBlock b;
            b.type = trim(line.substr(1, line.find(']') - 1));
            out.push_back(b);
        } else {
            size_t eq = line.find('=');
            BP_CHECK(eq != std::string::npos && !out.empty(), "malformed cfg line");
            out.back().kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
        }
    }
    // a leading [net]/[network] block (Darknet-C cfgs) carries no layer
    if (!out.empty() && (out[0].type == "net" || out[0].type == "network")) out.erase(out.begin());
    return out;
}
static std::vector<int> parse_ints(const std::string& s) {
    std::vector<int> v;
    std::istringstream is(s);
    std::string tok;
    while (std::getline(is, tok, ',')) {
        tok = trim(tok);
        if (!tok.empty()) v.push_back(std::atoi(tok.c_str()));
    }
    return v;
}

// ------------------------------------------------------------------ YoloNet
YoloNet::YoloNet(const std::string& cfg_text, const float* stream, size_t n_floats, int reso, int max_batch,
                 std::shared_ptr<WeightStore> store, bool darknet_bn)
    : Net(max_batch, store), cfg_text_(cfg_text), n_floats_(n_floats), reso_(reso) {
    darknet_bn_ = darknet_bn;
    BP_CHECK(reso % 32 == 0 && reso > 32, "reso must be a multiple of 32 and > 32 (dataloader.py:298-299)");
    const std::vector<CfgBlock> L = parse_cfg(cfg_text);
    const int n = (int)L.size();
    BP_CHECK(n > 0, "empty cfg");
    struct Shape { int C, H, W; };
    std::vector<Shape> shp(n);
    std::vector<int> alias(n);            // root layer whose tensor this layer's output is
    std::vector<std::vector<int>> members(n);
    // ---- shapes + aliases
    for (int i = 0; i < n; ++i) {
        const CfgBlock& b = L[i];
        Shape prev = i ? shp[i - 1] : Shape{3, reso, reso};
        alias[i] = i;
        if (b.type == "convolutional") {
            const int k = b.geti("size", 1), st = b.geti("stride", 1);
            BP_CHECK(k >= 1 && k <= 15 && st >= 1 && b.geti("filters", 1) >= 1, "convolutional: size / stride / filters out of range");
            const int pad = b.has("pad") && !b.gets("pad").empty() ? (k - 1) / 2 : 0;   // string truthiness, darknet.py:250
            BP_CHECK(prev.H + 2 * pad >= k && prev.W + 2 * pad >= k, "convolutional: kernel larger than its input");
            shp[i] = {b.geti("filters", 1), (prev.H + 2 * pad - k) / st + 1, (prev.W + 2 * pad - k) / st + 1};
        } else if (b.type == "shortcut") {
            BP_CHECK(i >= 1, "shortcut at layer 0");
            const int src = i + b.geti("from", -3);          // always relative (darknet.py:338)
            BP_CHECK(src >= 0 && src < i, "shortcut source out of range");
            BP_CHECK(shp[src].C == prev.C && shp[src].H == prev.H && shp[src].W == prev.W, "shortcut shape mismatch");
            shp[i] = prev;
        } else if (b.type == "upsample") {
            BP_CHECK(b.geti("stride", 2) == 2, "only x2 upsample");
            shp[i] = {prev.C, prev.H * 2, prev.W * 2};
        } else if (b.type == "route") {
            std::vector<int> ls = parse_ints(b.gets("layers"));
            BP_CHECK(ls.size() == 1 || ls.size() == 2, "route with 1 or 2 layers");
            const int a = i + ls[0];                       // always relative (darknet.py:347,352)
            BP_CHECK(a >= 0 && a < i, "route source");
            if (ls.size() == 1) {
                shp[i] = shp[a];
                alias[i] = alias[a];
            } else {
                const int c = ls[1];                       // absolute (darknet.py:351)
                BP_CHECK(c >= 0 && c < i, "route source");
                BP_CHECK(shp[a].H == shp[c].H && shp[a].W == shp[c].W, "route spatial mismatch");
                shp[i] = {shp[a].C + shp[c].C, shp[a].H, shp[a].W};
                members[i] = {alias[a], alias[c]};
            }
        } else if (b.type == "yolo") {
            BP_CHECK(i >= 1, "yolo at layer 0");
            shp[i] = prev;
            alias[i] = alias[i - 1];                        // outputs[i] = outputs[i-1]
        } else {
            throw Error("unsupported cfg block [" + b.type + "]");
        }
    }
    // ---- consumers per root
    std::vector<std::vector<int>> cons(n);
    auto use = [&](int src, int by) { cons[alias[src]].push_back(by); };
    for (int i = 0; i < n; ++i) {
        const CfgBlock& b = L[i];
        if (b.type == "convolutional" || b.type == "upsample") { if (i) use(i - 1, i); }
        else if (b.type == "shortcut") { use(i - 1, i); use(i + b.geti("from", -3), i); }
        else if (b.type == "route") { for (int m : members[i]) cons[m].push_back(i); if (members[i].empty()) {} }
        else if (b.type == "yolo") use(i - 1, i);
    }
    // ---- fusion: conv -> (shortcut | upsample) when the conv output has no other reader
    std::vector<int> fused_into(n, -1), fused_from(n, -1);
    for (int i = 0; i + 1 < n; ++i) {
        if (L[i].type != "convolutional") continue;
        const std::string& nt = L[i + 1].type;
        if (cons[i].size() == 1 && cons[i][0] == i + 1) {
            if (nt == "shortcut" && alias[i + 1 + L[i + 1].geti("from", -3)] != i) { fused_into[i] = i + 1; fused_from[i + 1] = i; }
            else if (nt == "upsample") { fused_into[i] = i + 1; fused_from[i + 1] = i; }
        }
    }
    // ---- tensors: concat buffers first, members become views
    std::vector<Tensor> T(n);
    std::vector<char> has(n, 0);
    in_nhwc_ = arena_.alloc((size_t)max_batch * reso * reso * 3);
    std::vector<std::pair<int, int>> copy_members;   // (route, member) pairs that need a copy op
    for (int i = 0; i < n; ++i) {
        if (members[i].empty()) continue;
        T[i] = new_tensor(shp[i].H, shp[i].W, shp[i].C);
        has[i] = 1;
        int off = 0;
        for (int m : members[i]) {
            const bool producible = L[m].type == "convolutional" || L[m].type == "shortcut" || L[m].type == "upsample";
            if (!has[m] && producible && members[m].empty()) {
                T[m] = T[i];
                T[m].p = T[i].p + off;
                T[m].C = shp[m].C;
                has[m] = 1;
            } else {
                copy_members.push_back({i, m});
            }
            off += shp[m].C;
        }
    }
    for (int i = 0; i < n; ++i) {
        if (has[i] || alias[i] != i) continue;
        if (L[i].type == "convolutional" && fused_into[i] >= 0) continue;   // lives only inside the fused epilogue
        T[i] = new_tensor(shp[i].H, shp[i].W, shp[i].C);
        has[i] = 1;
    }
    auto tensor_of = [&](int i) -> const Tensor& {
        const int r = alias[i];
        BP_CHECK(has[r], "internal: tensor not materialised");
        return T[r];
    };
    // ---- weights cursor
    size_t cur = 0;
    auto take = [&](size_t cnt) {
        BP_CHECK(cur + cnt <= n_floats, "weights stream too short for this cfg");
        const float* p = stream + cur;
        cur += cnt;
        return p;
    };
    // ---- emit
    Tensor input;
    input.p = in_nhwc_; input.H = reso; input.W = reso; input.C = 3; input.ld = 3;
    int row_off = 0;
    for (int i = 0; i < n; ++i) {
        const CfgBlock& b = L[i];
        if (b.type == "convolutional") {
            const Tensor& in = i ? tensor_of(i - 1) : input;
            const int k = b.geti("size", 1), st = b.geti("stride", 1);
            const int pad = b.has("pad") && !b.gets("pad").empty() ? (k - 1) / 2 : 0;
            const int Cout = shp[i].C, Cin = in.C;
            ConvWeights cw;
            if (b.geti("batch_normalize", 0) > 0) {
                cw.bn_bias = take(Cout); cw.bn_scale = take(Cout); cw.bn_mean = take(Cout); cw.bn_var = take(Cout);
            } else {
                cw.bias = take(Cout);
            }
            cw.w = take((size_t)Cout * Cin * k * k);
            const std::string a = b.gets("activation");
            const int act = a == "leaky" ? ACT_LEAKY : (a == "relu" ? ACT_RELU : ACT_LINEAR);
            BP_CHECK(a == "leaky" || a == "linear" || a == "relu", "unsupported activation " + a);
            int mode = ST_NHWC;
            const Tensor* res = nullptr;
            int dst = i;
            if (fused_into[i] >= 0) {
                dst = fused_into[i];
                if (L[dst].type == "shortcut") res = &tensor_of(dst + L[dst].geti("from", -3));
                else mode = ST_UP2;
            }
            add_conv("conv" + std::to_string(i), in, T[alias[dst]], cw, Cout, k, st, pad, act, mode, res, nullptr,
                     /*res_after_act=*/1, 1e-5f, shp[i].H, shp[i].W);
            add_tap(std::to_string(dst), T[alias[dst]]);
        } else if (b.type == "shortcut") {
            if (fused_from[i] >= 0) continue;
            const Tensor& a = tensor_of(i - 1);
            const Tensor& c = tensor_of(i + b.geti("from", -3));
            Op op; op.type = OP_ADD; op.name = "shortcut" + std::to_string(i);
            op.a = a.p; op.a_ld = a.ld; op.b = c.p; op.b_ld = c.ld; op.out = T[i].p; op.out_ld = T[i].ld;
            op.H = shp[i].H; op.W = shp[i].W; op.C = shp[i].C;
            ops_.push_back(op);
            add_tap(std::to_string(i), T[i]);
        } else if (b.type == "upsample") {
            if (fused_from[i] >= 0) continue;
            const Tensor& a = tensor_of(i - 1);
            Op op; op.type = OP_UPSAMPLE; op.name = "upsample" + std::to_string(i);
            op.a = a.p; op.a_ld = a.ld; op.out = T[i].p; op.out_ld = T[i].ld;
            op.H = a.H; op.W = a.W; op.C = a.C;
            ops_.push_back(op);
            add_tap(std::to_string(i), T[i]);
        } else if (b.type == "route") {
            if (members[i].empty()) continue;
            int off = 0;
            for (int m : members[i]) {
                for (auto& cm : copy_members)
                    if (cm.first == i && cm.second == m) {
                        Op op; op.type = OP_COPYCH; op.name = "concat" + std::to_string(i);
                        op.a = T[m].p; op.a_ld = T[m].ld; op.out = T[i].p + off; op.out_ld = T[i].ld;
                        op.H = shp[i].H; op.W = shp[i].W; op.C = shp[m].C;
                        ops_.push_back(op);
                    }
                off += shp[m].C;
            }
            add_tap(std::to_string(i), T[i]);
        } else if (b.type == "yolo") {
            const Tensor& t = tensor_of(i - 1);
            std::vector<int> mask = parse_ints(b.gets("mask"));
            std::vector<int> an = parse_ints(b.gets("anchors"));
            const int classes = b.geti("classes", 1);
            BP_CHECK(mask.size() == 3, "yolo layer needs 3 masked anchors");
            BP_CHECK(t.C == 3 * (5 + classes) && t.ld == t.C, "yolo head channel count");
            BP_CHECK(attrs_ == 0 || attrs_ == 5 + classes, "heads disagree on classes");
            attrs_ = 5 + classes;
            YoloHead h{};
            h.t = t.p; h.g = t.H; h.row_off = row_off;
            for (int a = 0; a < 3; ++a) {
                BP_CHECK(mask[a] >= 0 && 2 * mask[a] + 1 < (int)an.size(), "anchor mask");
                h.aw[a] = (float)an[2 * mask[a]];
                h.ah[a] = (float)an[2 * mask[a] + 1];
            }
            heads_.push_back(h);
            row_off += 3 * t.H * t.W;
        }
    }
    BP_CHECK(!heads_.empty() && heads_.size() <= 4, "cfg must have 1..4 yolo layers");
    rows_ = row_off;
    pred_ = arena_.alloc((size_t)max_batch * rows_ * attrs_);
    finalize();
}

void YoloNet::forward(const float* d_img, bool nhwc_input, int batch, float* d_pred, float conf, int num_classes,
                      float* d_sel, hipStream_t s, int sel_ld) {
    BP_CHECK(batch >= 1 && batch <= max_batch_, "batch out of range");
    if (nhwc_input) {
        if (d_img != in_nhwc_)
            BP_HIP(hipMemcpyAsync(in_nhwc_, d_img, (size_t)batch * reso_ * reso_ * 3 * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
    } else {
        launch_nchw_to_nhwc(d_img, in_nhwc_, batch, 3, reso_, reso_, s);
    }
    run_ops(batch, s);
    // heads hold per-image strides for max_batch_ == layout of batch b (contiguous by image), so decode as is
    static const bool two_kernels = std::getenv("BP_NO_DECODE_FUSION") != nullptr;   // A/B runs, tests
    if (d_sel && !d_pred && !two_kernels) {
        // nobody reads the prediction tensor (the fused per-frame pipeline): decode + select in one launch, same record
        launch_yolo_decode_select(heads_.data(), (int)heads_.size(), batch, reso_, attrs_, rows_, conf, num_classes, d_sel, s, sel_ld);
    } else {
        float* pred = d_pred ? d_pred : pred_;
        launch_yolo_decode(heads_.data(), (int)heads_.size(), batch, reso_, attrs_, rows_, pred, s);
        if (d_sel) launch_yolo_select(pred, batch, rows_, attrs_, conf, num_classes, d_sel, s, sel_ld);
    }
    BP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ KpdNet (FastPose)
KpdNet::KpdNet(const float* stream, size_t n_floats, int n_classes, int max_batch, int inH, int inW,
               std::shared_ptr<WeightStore> store)
    : Net(max_batch, store), n_floats_(n_floats), n_classes_(n_classes), inH_(inH), inW_(inW) {
    BP_CHECK(inH % 32 == 0 && inW % 32 == 0, "KPD input must be a multiple of 32");
    BP_CHECK(n_classes >= 1, "n_classes");
    outC_ = std::min(n_classes, 50);   // InferenNet_fast narrows to the first 50 maps (main_fast_inference.py:44)
    size_t cur = 0;
    auto take = [&](size_t cnt) {
        BP_CHECK(cur + cnt <= n_floats, "KPD stream too short");
        const float* p = stream + cur;
        cur += cnt;
        return p;
    };
    auto take_conv_bn = [&](int cout, int cin, int k) {
        ConvWeights cw;
        cw.bn_bias = take(cout); cw.bn_scale = take(cout); cw.bn_mean = take(cout); cw.bn_var = take(cout);
        cw.w = take((size_t)cout * cin * k * k);
        return cw;
    };
    auto upload = [&](const float* h, size_t cnt) { return upload_weights(h, cnt); };
    in_nhwc_ = arena_.alloc((size_t)max_batch * inH * inW * 3);
    Tensor x;
    x.p = in_nhwc_; x.H = inH; x.W = inW; x.C = 3; x.ld = 3;
    // stem: 7x7/2 + BN + ReLU, max-pool 3/2/1   (SE_Resnet.py:54-58,71)
    {
        ConvWeights cw = take_conv_bn(64, 3, 7);
        Tensor t = new_tensor(inH / 2, inW / 2, 64);
        add_conv("stem", x, t, cw, 64, 7, 2, 3, ACT_RELU, ST_NHWC, nullptr, nullptr, 0, 1e-5f, t.H, t.W);
        Tensor p = new_tensor(inH / 4, inW / 4, 64);
        Op op; op.type = OP_MAXPOOL; op.name = "maxpool";
        op.a = t.p; op.out = p.p; op.H = t.H; op.W = t.W; op.C = 64; op.OH = p.H; op.OW = p.W;
        ops_.push_back(op);
        x = p;
        add_tap("stem", x);
    }
    const int planes_[4] = {64, 128, 256, 512}, nblocks_[4] = {3, 4, 23, 3}, strides_[4] = {1, 2, 2, 2};
    int inplanes = 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = planes_[li];
        for (int bi = 0; bi < nblocks_[li]; ++bi) {
            const int st = bi == 0 ? strides_[li] : 1;
            const bool first = bi == 0;
            const std::string nm = "preact.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            ConvWeights c1 = take_conv_bn(planes, inplanes, 1);
            ConvWeights c2 = take_conv_bn(planes, planes, 3);
            ConvWeights c3 = take_conv_bn(planes * 4, planes, 1);
            const int OH = x.H / st, OW = x.W / st;
            Tensor t1 = new_tensor(x.H, x.W, planes);
            add_conv(nm + ".conv1", x, t1, c1, planes, 1, 1, 0, ACT_RELU, ST_NHWC, nullptr, nullptr, 0, 1e-5f, x.H, x.W);
            Tensor t2 = new_tensor(OH, OW, planes);
            add_conv(nm + ".conv2", t1, t2, c2, planes, 3, st, 1, ACT_RELU, ST_NHWC, nullptr, nullptr, 0, 1e-5f, OH, OW);
            Tensor out = new_tensor(OH, OW, planes * 4);
            if (!first) {
                // out = relu(bn3(conv3) + x)   (SE_Resnet.py:31-40)
                add_conv(nm + ".conv3", t2, out, c3, planes * 4, 1, 1, 0, ACT_RELU, ST_NHWC, &x, nullptr, 0, 1e-5f, OH, OW);
            } else {
                // SE block: T = bn3(conv3); y = sigmoid(fc2(relu(fc0(avgpool(T))))); out = relu(bn_d(conv_d(x)) + T*y)
                const int C = planes * 4;
                const float* w0 = take((size_t)C * C); const float* b0 = take(C);
                const float* w2 = take((size_t)C * C); const float* b2 = take(C);
                ConvWeights cd = take_conv_bn(C, inplanes, 1);
                Tensor Tt = new_tensor(OH, OW, C);
                add_conv(nm + ".conv3", t2, Tt, c3, C, 1, 1, 0, ACT_LINEAR, ST_NHWC, nullptr, nullptr, 0, 1e-5f, OH, OW);
                const int parts = avgpool_parts(OH * OW);
                const int tpi = (OH * OW + 63) / 64;      // slice sums per image when the pool rides in conv3's epilogue (pool_in_epilogue)
                float* pooled = arena_.alloc((size_t)max_batch * std::max(parts, tpi) * C);
                ops_.back().pool_out = pooled;
                float* hid = arena_.alloc((size_t)max_batch * C);
                float* y = arena_.alloc((size_t)max_batch * C);
                Op ap; ap.type = OP_AVGPOOL; ap.name = nm + ".se.pool";
                ap.a = Tt.p; ap.a_ld = Tt.ld; ap.out = pooled; ap.H = OH; ap.W = OW; ap.C = C;
                ops_.push_back(ap);
                Op f0; f0.type = OP_FC; f0.name = nm + ".se.fc.0";
                f0.a = pooled; f0.w = upload(w0, (size_t)C * C); f0.bias = upload(b0, C); f0.out = hid; f0.Cin = C; f0.Cout = C; f0.act = 2;
                f0.in_parts = parts; f0.in_scale = 1.f / (float)(OH * OW);
                f0.flops = 2.0 * C * C; f0.bytes = 4.0 * C * C;
                ops_.push_back(f0);
                Op f2 = f0; f2.name = nm + ".se.fc.2";
                f2.a = hid; f2.w = upload(w2, (size_t)C * C); f2.bias = upload(b2, C); f2.out = y; f2.act = 3;
                f2.in_parts = 1; f2.in_scale = 1.f;
                ops_.push_back(f2);
                add_conv(nm + ".downsample", x, out, cd, C, 1, st, 0, ACT_RELU, ST_NHWC, &Tt, y, 0, 1e-5f, OH, OW);
            }
            x = out;
            inplanes = planes * 4;
            add_tap(nm, x);
        }
    }
    // PixelShuffle(2): [2048, H/32, W/32] -> [512, H/16, W/16]   (FastPose.py:30)
    {
        Tensor t = new_tensor(x.H * 2, x.W * 2, x.C / 4);
        Op op; op.type = OP_PIXSHUF; op.name = "suffle1";
        op.a = x.p; op.out = t.p; op.H = x.H; op.W = x.W; op.C = x.C;
        ops_.push_back(op);
        x = t;
    }
    // DUC x2: conv3x3 + BN + ReLU with the PixelShuffle folded into the store   (DUC.py:18-23)
    const int duc_out[2] = {1024, 512};
    for (int d = 0; d < 2; ++d) {
        ConvWeights cw = take_conv_bn(duc_out[d], x.C, 3);
        Tensor t = new_tensor(x.H * 2, x.W * 2, duc_out[d] / 4);
        add_conv("duc" + std::to_string(d + 1), x, t, cw, duc_out[d], 3, 1, 1, ACT_RELU, ST_PIXSHUF, nullptr, nullptr, 0,
                 1e-5f, x.H, x.W);
        x = t;
        add_tap("duc" + std::to_string(d + 1), x);
    }
    // conv_out 3x3 (+bias), only the maps InferenNet_fast keeps, stored NCHW for the arg-max kernel
    {
        ConvWeights cw;
        cw.bias = take(n_classes);
        cw.w = take((size_t)n_classes * x.C * 9);
        hm_ = arena_.alloc((size_t)max_batch * outC_ * x.H * x.W);
        Tensor t; t.p = hm_; t.H = x.H; t.W = x.W; t.C = outC_; t.ld = outC_;
        hm_op_ = add_conv("conv_out", x, t, cw, outC_, 3, 1, 1, ACT_LINEAR, ST_NCHW, nullptr, nullptr, 0, 1e-5f, x.H, x.W);
    }
    BP_CHECK(cur == n_floats, "KPD stream has trailing data (wrong n_classes?)");
    finalize();
}

void KpdNet::forward(const float* d_inps, bool nhwc_input, int batch, float* d_hm, float* d_kp, hipStream_t s, int kp_ld) {
    BP_CHECK(batch >= 1 && batch <= max_batch_, "batch out of range");
    if (nhwc_input) {
        if (d_inps != in_nhwc_)
            BP_HIP(hipMemcpyAsync(in_nhwc_, d_inps, (size_t)batch * inH_ * inW_ * 3 * sizeof(float),
                                  hipMemcpyDeviceToDevice, s));
    } else {
        launch_nchw_to_nhwc(d_inps, in_nhwc_, batch, 3, inH_, inW_, s);
    }
    // the head writes straight into the caller's tensor -- for THIS pass only: a pointer left bound would be written by the next profile() /
    // tap pass at whatever batch size that one runs (round 6: profile(28) behind a one-crop forward wrote 28 maps into a one-map tensor)
    struct Rebind { float*& slot; float* own; ~Rebind() { slot = own; } } rebind{ops_[hm_op_].conv.out, hm_};
    ops_[hm_op_].conv.out = d_hm ? d_hm : hm_;
    run_ops(batch, s);
    if (d_kp) launch_heatmap_argmax(d_hm ? d_hm : hm_, batch, outC_, out_h(), out_w(), d_kp, s, kp_ld);
    BP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ Pillow bicubic coefficient tables
// Restates Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c; third-party,
// not under /root/reference; pinned integer-exactly against the installed Pillow in the tests).
static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
ResizePlan make_bicubic_plan(int in_size, int out_size) {
    ResizePlan pl;
    pl.in_size = in_size; pl.out_size = out_size;
    const double scale = (double)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    pl.ksize = ksize;
    pl.bounds.assign((size_t)out_size * 2, 0);
    pl.coeffs.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < xmax; ++x) {
            const double v = k[x] * (double)(1 << 22);
            pl.coeffs[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        pl.bounds[2 * xx] = xmin;
        pl.bounds[2 * xx + 1] = xmax;
    }
    return pl;
}

}  // namespace bp
