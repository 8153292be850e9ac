// planner.h -- host-side level planner of the gen.phi hot path (no HIP here).
//
// Turns the rank-ordered pedigree + proband list into a sequence of "level steps".
// Step s computes the kinship matrix of cut s+1 from the matrix of cut s; what the
// reference does with IndexedIndividual objects, founder_index and a recursive per-pair
// kernel (src/compute.jl:105-158, :233-304) becomes flat int32 index arrays:
//
//   every member x of cut s+1 has up to two SOURCES in cut s
//       new, both parents : A = father, B = mother        weight 1/2
//       new, one parent   : A = that parent, B = none     weight 1/2
//       new, no parent    : A = none, B = none            weight 1/2 (row/col of zeros)
//       dragged (x is also in cut s) : A = x itself       weight 1
//   and   phi[i][j] = w_i * w_j * sum_{p in src(i), q in src(j)} Psi[p][q]
//   (Float64, grouped as the reference groups it), except the diagonal of a
//   new individual, which is 1/2 + Psi[A][B]/2.
//
// "none" is encoded as index n_prev: every level matrix carries one extra all-zero row
// and zero columns up to its pitch, so kernels gather unconditionally.
//
// Storage order of a cut (free for intermediate cuts):
//   [dragged members by their position in the previous cut ..., new members by rank ...]
// so that (i) "row member has the larger rank" is a position test among new members, (ii) the
// sources of the dragged columns increase monotonically (a level's dragged x dragged block is a
// stream compaction of the previous matrix), (iii) the new x new block is contiguous.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace genphi {

constexpr int32_t kNewFlag = INT32_MIN;     // bit 31 of `ord`: member is new (weight 1/2)

// Kernel family of a level step, by the width of the previous cut:
//   FULL   both source rows of an output row fit in LDS
//   SPLIT  one source row at a time fits (persistent pipelined kernels, certified fast path)
//   WIDE   a source row does not fit in LDS (or indices need more than 16 bits): the level is
//          assembled from streaming passes -- dragged x dragged and new x dragged blocks by row
//          compaction, dragged x new by transposition, new x new by a FULL / SPLIT sub-step on
//          the compacted parent x parent matrix (LevelStep::parents / nn)
enum { kModeFull = 0, kModeSplit = 1, kModeWide = 2 };

struct LevelStep {
    int64_t n_prev = 0, n = 0;        // |cut s|, |cut s+1|
    int64_t ld_prev = 0, ld = 0;      // row pitch (floats) of the two level matrices
    int64_t width = 0;                // columns a row kernel writes: [0, n) and the zero padding [n, width); = ld except in a sub-step
    int64_t n_dragged = 0;            // |cut s ∩ cut s+1|  ("both" in the verbose lines)
    // per member of cut s+1, in this cut's storage order:
    std::vector<int32_t> srcA, srcB;  // positions in cut s; n_prev = none
    std::vector<int32_t> ord;         // pedigree rank index (0-based) | kNewFlag
    std::vector<int32_t> work;        // row processing order (rows sharing srcA adjacent)
    std::vector<uint32_t> pk;         // srcA | srcB << 16 (FULL / SPLIT modes)
    int mode = 0;
    bool pos_ord = false;             // new members with both parents appear in rank order along the
                                      // storage order: kernels use the position test instead of rank words
    // WIDE only.  The cut is stored [dragged..., new...] (n_dragged, n - n_dragged).
    std::vector<int32_t> parents;     // distinct parents of the new members: positions in cut s, ascending
    // the new x new block as a level step of its own over Psi_P = Psi[parents][parents]:
    // n_prev = |parents|, sources = indices into `parents` (|parents| = none); mode FULL or SPLIT
    // (0 or 1 element; empty with nn_naive when |parents| is too wide even for SPLIT: the block then
    // comes from the per-entry kernel on Psi itself).  The sub-step writes the block IN PLACE, into
    // rows / columns [n_dragged, n) of the cut's matrix: its pitch is the cut's, and it starts with
    // lead = n_dragged % 4 placeholder members (no sources, not in `work`) so that its column 0 sits
    // on a 16-byte boundary of the cut's rows: n = lead + n_new, width = ld - (n_dragged - lead).
    std::vector<LevelStep> nn;
    int32_t lead = 0;                 // (of a sub-step) placeholder members in front
    bool nn_naive = false;
    // WIDE, members that STAY where they are (persistent slots; Float32 sweep only -- everything above keeps its compact meaning
    // for the other sweeps).  A run of consecutive WIDE steps can keep ONE level matrix of pitch / capacity P in place: a member
    // owns the same row and column (its SLOT, an absolute position in [0, P)) for as long as it is in the cuts, so the
    // dragged x dragged block -- most of a level of overlapping generations -- is never copied.  New members take free GRANULES
    // of 64 slots (everything in them left the cuts at least one step earlier), in the order of their block: new member r sits
    // at blk_slot[r / 64] + r % 64.  Inside a block members are ordered by the step at which they leave (earliest first), so
    // that what is born together and dies together frees whole granules.
    //   src_slots   the source cut is stored by slot: absA / absB = the sources' slots (P = none), parents_abs likewise
    //   stay        the output cut is stored by slot, in the SAME matrix: only the new rows / columns are written (npad =
    //               64 * granules >= n_new slots reserved; p0 = the first granule, diagnostic)
    //   out_slots   slot of every member of the output cut (stay steps)
    bool src_slots = false, stay = false;
    bool contig = false;              // (stay) the new members' slots are ONE stretch [p0, p0 + npad): the new x new block is written in place
    int32_t P = 0, p0 = 0, npad = 0;
    std::vector<int32_t> absA, absB, parents_abs, out_slots, blk_slot;
    std::vector<int32_t> live_ranges;      // (stay) [lo, hi) slot ranges, 64-aligned, ascending, that hold the dragged members
};

struct Plan {
    int64_t n_ind = 0;
    int64_t n_pro = 0;                       // distinct probands = side of the result
    int32_t n_levels = 0;                    // L (number of cuts)
    std::vector<int64_t> cut_sizes;          // L entries, top founders first
    std::vector<int64_t> both_counts;        // L-1 entries
    std::vector<int64_t> ld;                 // row pitch of each cut's matrix
    std::vector<LevelStep> steps;            // L-1 entries
    std::vector<int32_t> final_members;      // rank index of each proband, result order
    // The final cut is kept in [dragged, new] order when its step is WIDE; perm maps result
    // (proband) position -> storage position.  Empty = storage order is proband order.
    std::vector<int32_t> final_perm;
    // The last cut stayed in place (LevelStep::stay of the last step): slot of every proband, result order, in the run's matrix.  The
    // result is then delivered by ONE permutation pass from that matrix (its pitch -- ld[L-1] -- is the run's); final_perm keeps the
    // compact [dragged, new] meaning for the sweeps that know no slots.
    std::vector<int32_t> final_slots;
    double algorithmic_bytes = 0.0;          // 4 * sum (n_k^2 + n_{k+1}^2)
    int64_t max_cut = 0;
};

struct PlanOptions {
    int32_t full_max_floats = 8192;    // rows longer than this use the pipelined SPLIT kernel even when two rows would fit (tuned on genea140 / cfg3)
    int32_t lds_cap_floats = 36864;   // floats of LDS a workgroup may use for staged source rows (9 * 1024 float4; 160 KB minus the work-queue slots)
    bool indices_only = false;        // cuts and per-member sources / rank words only (every step marked FULL, no pk words, no
                                      // work order, the last cut in proband order): what the column-panel multi-GPU path needs
    bool no_stay = false;             // never keep WIDE levels in place (A/B and test hook)
    bool stay_scatter = false;        // in-place steps never write their new x new block in place (A/B and test hook: always through the compact buffer)
    int64_t stay_max_slots = 200000;  // largest slot capacity P of a run (a P x P Float32 matrix: 160 GB)
    double stay_mem_ratio = 1.2;      // in-place runs are dropped when the two level buffers would need more than this x plain alternation
    double stay_mem_floor_bytes = 4294967296.0;   // ... and more than this many bytes (4 GiB: the slot matrices of narrow cuts are small change)
    int32_t stay_min_ratio_pct = 200; // a step stays in place only while its cut has at least this % of its new members (few dragged members: little to save)
    int32_t stay_slack_pct = 6;       // free slots a run starts with beyond its widest (cut + new members), in % (granules that are only partly dead)
    bool stay_narrow = true;          // cuts whose rows fit in LDS (FULL / SPLIT widths) may stay in place too: their steps are switched to block
                                      // assembly when the cost model says the dragged x dragged copy that is saved outweighs the extra passes
    double stay_step_overhead = 64e6;  // fixed cost of a block-assembled step in the cost model, in matrix entries (its six to eight short launches; a row-kernel
                                      // level is charged 15 % of it); tests that put tiny cuts in place set it to 0
    bool stay_last = true;            // the proband cut may stay in place at the end of a run (the result is delivered from the slot matrix)
    bool stay_narrow_force = false;   // A/B hook: ... whatever the cost model says (every step that passes the ratio test)
    int64_t stay_narrow_min = 2048;   // ... from this width of the source cut on (narrower levels are bound by their launches, not their bytes)
    bool stay_family_order = true;    // new members of a cut of an in-place plan: siblings (same father) adjacent inside a leaving class
    int32_t stay_headroom = 0;        // extra blocks of free slots a run starts with (each the size of its largest block of new members):
                                      // more of them = longer runs before the slot space is full (memory: P grows)
};

// Returns 0 or a GENPHI_ERR_* code (see include/genphi.h); message in err.
int build_plan(int64_t n_ind, const int64_t *ind, const int64_t *father, const int64_t *mother,
               int64_t n_pro, const int64_t *pro_ids, const PlanOptions &opt, Plan &plan,
               std::string &err);

// Reorders a list of storage rows of cut s+1 for source-row reuse: rows sharing the A source
// stay adjacent (sibling groups), and groups are visited depth-first along shared B sources, so
// that the second use of a B (mother) row follows the first closely enough to be served by
// L2 / the Infinity Cache instead of HBM.
void reuse_order(const LevelStep &step, std::vector<int32_t> &rows);

// Work lists of a SPLIT launch: a HUB WALK over the parent graph of the launch's rows.
// A row with two sources (A, B) is an edge between the rows A and B of the previous level.  A workgroup that
// holds the expansion of a "hub" row H in registers (per column j: (Psi[H][A_j] + Psi[H][B_j]) / 4) finishes
//   - the rows whose only source is H (the dragged member H itself, one-parent children) from it alone,
//   - every child (H, X) by staging row X:  out = RN32(expansion(H) + expansion(X)),
// and the expansion of the LAST such X is a by-product of its stage, so the walk goes on with X as the hub
// without staging a hub row again.  A RUN is such a walk (an item of the kernels' work queue), cut into SEGMENTS
// of one hub and at most `seg_cap` children with a B source:
//   desc4[4 w ..]  = (storage row, output row, B source (= the row to stage; none: finished from the hub alone), rank word)
//   seg4[4 g ..]   = (first work row, hub row, number of leading work rows without B source, type)  + a terminator
//                    type 0: the run starts here (the hub row is staged), 1: the hub is the B row of the previous
//                    segment's last child (its expansion is already in registers), 2: same hub as the previous segment
//   run[4 r ..]    = (first segment of run r, its hub row | its n0 << 16, its first work row, its end work row)  + a terminator
//                    (the first segment again, so that ONE load starts an item; hub rows are < 65536 in SPLIT steps)
// Random mating (every member about two children): ~16 % fewer staged rows per level than one group per father;
// the 1e5-wide last level of cfg4: ~12 % fewer.  The order of the start rows follows `rows` (the planner's reuse order).
struct WalkLists {
    std::vector<int32_t> desc4, seg4, run;
    std::vector<int32_t> row_k;        // per work row: its index in `rows`
};
void build_hub_walk(const int32_t *srcA, const int32_t *srcB, const int32_t *ord, int32_t none, const int *rows,
                    const int *out_rows, int n_rows, int seg_cap, int max_run, WalkLists &out);

inline int64_t pitch_for(int64_t n) { return ((n + 1) + 63) / 64 * 64; }

// Tuning, A/B and test hooks are GENPHI_* environment variables that the library reads ONLY when GENPHI_ENV_HOOKS=1 is set as
// well: a shared library loaded into somebody's Julia process does not change kernels on ambient variables.  (Programmatic
// settings: genphi_tuning, include/genphi.h.  Not gated: GENPHI_TRACE -- diagnostics on stderr -- and the memory budgets
// GENPHI_KEEP_MB / GENPHI_SPARSE_KEEP_MB, which change no result and no kernel.)
inline const char *env_hook(const char *name)
{
    static const bool on = [] { const char *e = std::getenv("GENPHI_ENV_HOOKS"); return e && std::atoi(e) != 0; }();
    return on ? std::getenv(name) : nullptr;
}

// GENPHI_TRACE=1: wall-clock marks of the phases of a call on stderr (planner, upload, sweep): where does the host side of a call go?
struct PhaseTrace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    PhaseTrace() : on(std::getenv("GENPHI_TRACE") != nullptr), t0(std::chrono::steady_clock::now()), last(t0) {}
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[genphi trace] %-28s +%8.3f ms  (at %8.3f ms)\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
                     std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    }
};

}  // namespace genphi
