// panel_launch.h -- internal interface between the column-panel path (panel_phi.hip) and the row kernels
// of genphi_hip.hip: a level step of a rank's panel runs the same FULL / SPLIT kernels as a whole level,
// on the rank's local columns and with rows of the rank's extended panel as source rows.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace genphi {

constexpr size_t kPanelIdxPad = 32 * 1024;     // padding entries behind pk_col / ord_col (the row kernels read whole quads past the end)
constexpr int kPanelFullMaxFloats = 8192;      // panel rows up to this many floats: FULL (both source rows in LDS)
constexpr int kPanelSplitMaxFloats = 36864;    // ... up to this: SPLIT (one source row in LDS); beyond: per-entry kernel

struct PanelLaunch {
    hipStream_t stream;
    int n_cus;
    const void *tuning;          // panel_tuning_create()
    const float *psi;            // extended local panel of the previous cut: (n_prev + 1) rows x ld_prev
    float *out;                  // local panel of this cut: (n_cut + 1) rows x ld
    long long ld_prev, ld;
    int n_prev, n_cut;           // members of the previous / this cut (= row index of the zero row of psi / out)
    int n_cols;                  // local columns of this cut
    int src_width;               // floats of a source panel row including its zero column
    const int *srcA, *srcB, *ord;    // per ROW of this cut: positions in the previous cut (n_prev = none), rank word
    const unsigned *pk_col;      // per local column: panel column of its A source | of its B source << 16 (padded)
    const int *ord_col;          // per local column: rank word (padded)
    const int *diag_col;         // per row: its local column, or -1
    const int *work;             // rows in work order (rows sharing the A source adjacent), n_cut entries
    int mode;                    // 0 FULL, 1 SPLIT
    const int4 *desc;            // SPLIT: the hub walk of the cut's rows (WalkLists, planner.h): per work row (row, row, B source, rank word)
    const int4 *seg;             // SPLIT: per segment (first work row, hub row, leading rows without B source, type) + terminators
    const int4 *run;             // SPLIT: (first segment, hub row | n0 << 16, its rows [z, w)) of every run + terminator
    const int2 *pdesc;           // SPLIT: per work row (local column or -1, panel column of the member's OTHER source: the hub of its segment)
    int n_segs, n_runs;
    const int *cert_prev;        // exactness certificates of the rows of psi (this rank's columns of them)
    int *cert_out;               // ... of the rows written (zeroed by the caller)
    int *counters;               // 20 ints (work queues + group counts), zeroed by launch_panel_level
    int *glist;                  // 2 * glist_cap ints
    int glist_cap;
};

const void *panel_tuning_create();               // environment hooks, read once per panel handle
void panel_tuning_destroy(const void *t);
int panel_tuning_lds_cap(const void *t, int dflt);        // GENPHI_LDS_CAP_FLOATS (>= 16) or dflt
int panel_tuning_full_max(const void *t, int dflt);       // GENPHI_FULL_MAX_FLOATS (>= 0) or dflt
unsigned panel_tuning_cert_thresh(const void *t);         // certificate threshold word (GENPHI_CERT_MIN_EXP raises it)
int launch_panel_level(const PanelLaunch &L);    // GENPHI_OK or an error code (message in genphi_last_error)

}  // namespace genphi
