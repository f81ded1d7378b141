// vibo_capi.hip -- C ABI of libvibo_hip.so (see include/vibo_hip.h) plus the small
// helper kernels around the fused ELBO kernel: item prep, partial finalize,
// forward-only encode, decode.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vibo_hip.h"
#include "vibo_device.hpp"
#include "vibo_cond.hpp"
#include "vibo_cond_finalize.hpp"
#include "vibo_finalize.hpp"
#include "vibo_general.hpp"
#include "vibo_launch.hpp"
#include "vibo_multi.hpp"
#include "vibo_params.hpp"
#include "vibo_train_hook.hpp"

namespace vibo {

static thread_local char g_err[512] = "";
// measurement hook (vibo_set_insitu_timer): the timer block the matrix row-split launches of THIS host thread stamp; null = none
static thread_local unsigned long long* g_insitu = nullptr;

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e > 0 ? (int)e : 999;
}

static bool rows_chunkable(const vibo_desc* d);
static int item_feat_dim(int irt, int A) { return irt == 1 ? 1 : (irt == 2 ? A + 1 : A + 2); }

static int check_desc(const vibo_desc* d) {
    if (!d) return fail(-1, "null descriptor");
    if (d->abi_version != VIBO_ABI_VERSION) return fail(-2, "abi_version %d != %d", d->abi_version, VIBO_ABI_VERSION);
    if (d->num_person < 1) return fail(-3, "num_person must be >= 1");
    if (d->num_item < 1) return fail(-3, "num_item must be >= 1");
    if (d->ability_dim < 1 || d->ability_dim > VIBO_MAX_ABILITY_DIM_WIDE)
        return fail(-3, "ability_dim %d outside 1..%d", d->ability_dim, VIBO_MAX_ABILITY_DIM_WIDE);
    if (d->ability_dim > VIBO_MAX_ABILITY_DIM && d->posterior == VIBO_POSTERIOR_GIVEN)
        return fail(-8, "VIBO_POSTERIOR_GIVEN needs the row-split path: ability_dim <= %d", VIBO_MAX_ABILITY_DIM);
    if (d->ability_dim > VIBO_MAX_ABILITY_DIM && (d->mask_dtype == VIBO_MASK_CODES))
        return fail(-8, "cell codes (VIBO_MASK_CODES) need the row-split paths: ability_dim <= %d", VIBO_MAX_ABILITY_DIM);
    if (d->irt_model < 1 || d->irt_model > 3) return fail(-3, "irt_model must be 1, 2 or 3");
    if (d->posterior != VIBO_POSTERIOR_UNCONDITIONAL && d->posterior != VIBO_POSTERIOR_CONDITIONAL &&
        d->posterior != VIBO_POSTERIOR_GIVEN)
        return fail(-3, "bad posterior");
    if (d->missing_mode != VIBO_MISSING_PRIOR && d->missing_mode != VIBO_MISSING_DROP) return fail(-3, "bad missing_mode");
    if (d->mask_dtype < 0 || d->mask_dtype > VIBO_MASK_CODES) return fail(-3, "bad mask_dtype");
    if (d->reg_mode != VIBO_REG_KL && d->reg_mode != VIBO_REG_SAMPLED) return fail(-3, "bad reg_mode");
    if (d->n_flows < 0 || d->n_flows > VIBO_MAX_FLOWS) return fail(-3, "n_flows outside 0..%d", VIBO_MAX_FLOWS);
    if (d->n_flows > 0 && d->reg_mode != VIBO_REG_SAMPLED) return fail(-3, "flows need reg_mode SAMPLED");
    if (d->flags & ~(VIBO_FLAG_KERNEL_VALU | VIBO_FLAG_KERNEL_MATRIX | VIBO_FLAG_NO_EMIT_CODES | VIBO_FLAG_COND_VALU | VIBO_FLAG_COND_MATRIX |
                     VIBO_FLAG_COND_THREE_PASS)) return fail(-3, "unknown flags");
    if ((d->flags & VIBO_FLAG_KERNEL_VALU) && (d->flags & VIBO_FLAG_KERNEL_MATRIX)) return fail(-3, "flags pin two kernels");
    if ((d->flags & VIBO_FLAG_COND_VALU) && (d->flags & VIBO_FLAG_COND_MATRIX)) return fail(-3, "flags pin two forms of the conditional passes");
    return 0;
}

struct Plan {
    bool general;             // wave-per-person kernel (conditional posterior / flows / > 1024 items)
    bool row_ok;              // wave-per-row register kernel is applicable (subject to alignment)
    int row_nblk;
    bool split_ok;            // row-split register kernel (ability_dim 3..8) is applicable (subject to alignment)
    int split_nq, split_nblk;
    int cond_nblk;            // workgroups of the conditional posterior's cond_pre launches (2 per CU)
    int cond_post_nblk;       // ... of cond_post: 3 per CU when it reads cell codes at template width <= 2 (its launch bound there)
    bool msplit;              // the row-split launches go to the matrix-pipe kernel (vibo_msplit_kernel.hpp): split_nq = waves of
                              // 128 items per workgroup, batches of 32 rows
    bool narrow;              // ... to the narrow-row kernel (vibo_narrow.hip: <= 128 items, a row per 16 lanes); split_nblk = its grid
    int panels;               // > 0: more than 1024 items, one row-split launch per panel of 1024 items
    size_t off_cnt;           // panel mode: per-person packed counts of the whole row
    bool cond;                // panel mode with the conditional posterior: cond_pre / split / cond_post per panel
    bool cmat_pre, cmat_post; // ... whose first / last pass runs on the matrix pipe from the cell codes, all items at once (vibo_cmean.hip)
    bool cond_fused;          // ... whose first pass is folded into the matrix row-split kernel (one panel, ability_dim 1, fp32 rows: its XM == 3)
    bool given;               // panel mode with a caller-supplied per-person posterior (VIBO_POSTERIOR_GIVEN)
    size_t off_pre, off_coef, off_cpart;
    size_t off_codes;         // fp32 rows read by more than one pass: the first pass's 1-byte cell codes [B][codes_stride] (0: not used)
    long long codes_stride;
    int cond_rec;             // floats per cond_post workgroup record
    int AT, D, DP, n_tiles, nblk, lds_main;
    LaunchGeom geom;
    PartialLayout lay;
    size_t off_item_prep, off_partial, total_bytes;
};

// compute units of the current device (asked per call: the library keeps no state between calls)
static int device_cus() {
    int dev = 0, n = 0;
    return (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
}

static hipError_t launch_split(const ElboParams& p, int AT, bool codes, int irt, bool grad, int nq, int grid, hipStream_t s,
                               bool msplit = false) {
    if (msplit) {
        const int nw = (p.I + 127) / 128;
        if (p.n_flows > 0) {
            if (codes) return launch_elbo_msplit_fc(p, irt, grad, nw, grid, s);
            if (p.row_index) return launch_elbo_msplit_fg(p, irt, grad, nw, grid, s);
            return launch_elbo_msplit_fa(p, irt, grad, nw, grid, s);
        }
        if (codes) return launch_elbo_msplit_c(p, irt, grad, nw, grid, s);
        if (p.row_index) return launch_elbo_msplit_g(p, irt, grad, nw, grid, s);
        return launch_elbo_msplit_a(p, irt, grad, nw, grid, s);
    }
    if (codes)
        return AT <= 2   ? launch_elbo_split_c2(p, irt, grad, nq, grid, s)
               : AT == 4 ? launch_elbo_split_c4(p, irt, grad, nq, grid, s)
                         : launch_elbo_split_c8(p, irt, grad, nq, grid, s);
    if (p.row_index)
        return AT <= 2   ? launch_elbo_split_g2(p, irt, grad, nq, grid, s)
               : AT == 4 ? launch_elbo_split_g4(p, irt, grad, nq, grid, s)
                         : launch_elbo_split_g8(p, irt, grad, nq, grid, s);
    return AT <= 2   ? launch_elbo_split_a2(p, irt, grad, nq, grid, s)
           : AT == 4 ? launch_elbo_split_a4(p, irt, grad, nq, grid, s)
                     : launch_elbo_split_a8(p, irt, grad, nq, grid, s);
}

// 16-byte row chunks need I % 4 == 0, or row strides that pad every row to a multiple of 4 cells (the cells past
// the row's end are read but masked out in the kernels)
static bool rows_chunkable(const vibo_desc* d) {
    const int I = d->num_item;
    if (I % 4 == 0) return true;
    const long long i4 = (I + 3) & ~3;
    if (d->mask_dtype != VIBO_MASK_CODES && d->response_row_stride < i4) return false;
    if ((d->mask_dtype == VIBO_MASK_U8 || d->mask_dtype == VIBO_MASK_CODES) && d->mask_row_stride < i4) return false;
    return true;
}

// rows can be read in aligned chunks of 4 cells (16 B of responses + 4 B of mask, or 4 B of cell codes)
static bool rows_vec_ok(const vibo_desc* d, const float* response, const void* mask) {
    bool vec = rows_chunkable(d);
    if (d->mask_dtype != VIBO_MASK_CODES) vec = vec && (d->response_row_stride % 4 == 0) && (((uintptr_t)response & 15) == 0);
    if (d->mask_dtype == VIBO_MASK_U8 || d->mask_dtype == VIBO_MASK_CODES)
        vec = vec && (d->mask_row_stride % 4 == 0) && (((uintptr_t)mask & 3) == 0);
    if (d->mask_dtype == VIBO_MASK_I64) vec = vec && (d->mask_row_stride % 2 == 0) && (((uintptr_t)mask & 15) == 0);
    return vec;
}
static int codes_unsupported() {
    return fail(-8, "cell codes (VIBO_MASK_CODES) need the row-split paths: 4..32767 items, rows 4-byte aligned with a "
                    "stride that pads them to a multiple of 4 cells");
}

// cell-code rows leave room for a third wave per SIMD in the narrower row-split kernels (see split_kernel's launch bounds)
static bool codes_three_waves(const vibo_desc* d, int AT) {
    return d->mask_dtype == VIBO_MASK_CODES && (AT <= 2 || (AT == 4 && d->irt_model <= 2));
}

// Which row-split kernel: the matrix-pipe kernel (contractions as f16 hi/lo MFMAs) or the VALU kernel.  The descriptor's
// flags pin one of them (A/B measurements, tests of both paths); VIBO_FLAG_NO_EMIT_CODES: later passes re-read the fp32 rows.
static bool emit_codes_wanted(const vibo_desc* d) { return !(d->flags & VIBO_FLAG_NO_EMIT_CODES); }
static bool want_msplit(const vibo_desc* d) {
    if (d->flags & VIBO_FLAG_KERNEL_VALU) return false;
    // (32-bit row numbers and batch counters in the matrix kernel)
    if (d->num_person > 0x7fffffff - 0x10000) return false;
    if (d->flags & VIBO_FLAG_KERNEL_MATRIX) return true;
    // Thresholds from tools/calibrate_planner.py (hipGraph replays of both kernels over persons x items x ability_dim on an
    // MI355X, profiles/r03_planner_calibration.txt):
    //  * small minibatches (the reference CLI's default is 16 persons): the matrix kernel's fixed cost -- operand images,
    //    512-thread workgroups, one batch of 32 rows per workgroup -- loses to the VALU kernel's 8-row batches: 1 000 items,
    //    ability_dim 8: 17 vs 22 us at 256 persons, 24 vs 24 at 2 048, 33 vs 27 at 4 096
    //  * narrow matrices: a workgroup of the matrix kernel is ceil(I / 128) waves on one CU, so with few items the chip holds
    //    few waves; the VALU kernel's 256-item waves and 2 workgroups per CU do better there
    //  * ability_dim <= 4: the contractions are a small part of the VALU kernel's work, the matrix kernel only wins once every
    //    workgroup streams several batches (65 536 x 1 000: 80 vs 91 us; 16 384 x 1 000: 36 vs 33)
    // Conditional posterior at ability_dim 1 on fp32 rows: the matrix kernel also replaces the first pass there (its XM == 3), which
    // moves the break-even down (tools/calibrate_planner.py-style hipGraph replays, round 6: 8 192 x 1 000 49 vs 57 us, 16 384 x 1 000
    // 59 vs 72, 65 536 x 512 111 vs 165, 65 536 x 256 102 vs 148; 4 096 persons: 42 vs 45 at 1 000 items, 41 vs 40 at 768, 35 vs 31 at 256)
    if (d->posterior == VIBO_POSTERIOR_CONDITIONAL && d->ability_dim == 1 && d->n_flows == 0 && d->num_item <= 1024 && d->num_item >= 256 &&
        d->mask_dtype != VIBO_MASK_CODES && d->mask_dtype != VIBO_MASK_I64 && !(d->flags & (VIBO_FLAG_COND_THREE_PASS | VIBO_FLAG_COND_VALU)) &&
        (!d->want_grad || emit_codes_wanted(d)))
        return d->num_person >= (d->num_item >= 896 ? 4096 : 8192);
    const int width = d->num_item < 1024 ? d->num_item : 1024;
    // (round 6, profiles/r06_planner_calibration.txt: the kernel's launch got ~9 us shorter -- 2 048 x 1 000 at ability_dim 8 19.4 vs 22.6 us,
    //  and the 385..512-item exclusion of round 3 -- "2 workgroups per CU with one batch each at 16 384 persons: 48 vs 42 us" -- now
    //  measures 18.2 vs 21.9 us at 4 096 persons, 36.4 vs 38.8 at 16 384: dropped)
    if (d->ability_dim >= 5 && width >= 896 && d->num_person >= 2048) return true;
    if (d->num_person < 4096) return false;
    const bool many = d->num_person >= 32768;
    if (d->ability_dim <= 4) return many && width >= 640;
    if (width < 320) return false;
    return true;
}
// Narrow rows (4..128 items: BASELINE configs[0] and [3]) of the plain model: the kernel that gives a row to 16 lanes instead of a
// whole wave (vibo_narrow.hip).  Either pinning flag keeps the row-split kernels (tests and A/B runs of those paths).
static bool want_narrow(const vibo_desc* d) {
    if (d->flags & (VIBO_FLAG_KERNEL_VALU | VIBO_FLAG_KERNEL_MATRIX)) return false;
    return d->num_item >= 4 && d->num_item <= 128 && d->ability_dim <= 4 && d->n_flows == 0 &&
           d->posterior == VIBO_POSTERIOR_UNCONDITIONAL && d->mask_dtype != VIBO_MASK_I64;
}
static int narrow_blocks(int num_cu, const vibo_desc* d) {
    // a workgroup = 4 waves = one per SIMD; workgroups per CU = the waves per SIMD the instantiation is compiled for
    // (narrow_waves_per_simd in vibo_narrow.hip); under 1024 records so that the fused train epilogue can finalize them
    const int il = d->num_item <= 64 ? 4 : 8, at = d->ability_dim <= 1 ? 1 : d->ability_dim <= 2 ? 2 : 4;
    const bool g3 = d->irt_model == 3 && d->want_grad;
    const int wps = narrow_waves_per_simd(at, il, g3);      // (vibo_launch.hpp: the kernel's launch bounds use the same function)
    long long nblk = (long long)num_cu * wps;
    if (nblk > 1020) nblk = 1020;
    const long long need = (d->num_person + 15) / 16;          // 4 rows per wave and round
    return (int)(nblk < need ? nblk : (need > 0 ? need : 1));
}
static int msplit_blocks(int num_cu, int items, long long persons) {
    const int nw = (items + 127) / 128;
    // workgroups per CU = what is resident at once: 2 waves per SIMD (the kernel's register budget) = 8 waves per CU, and the
    // 17.6 KB of LDS per wave stay under 160 KB with them.  (Round 2 launched 2 per CU at 5..7 waves and 4 at 3 waves: the
    // surplus workgroups queued behind the resident ones -- with one 32-row batch each that doubled the call:
    // 16 384 x 768 at ability_dim 8 57 us against the VALU kernel's 45, tools/calibrate_planner.py)
    long long nblk = (long long)num_cu * (8 / nw > 1 ? 8 / nw : 1);
    const long long nb = (persons + 31) / 32;
    return (int)(nblk < nb ? nblk : nb);
}

static int make_plan(const vibo_desc* d, Plan* pl, bool allow_msplit = true) {
    const int num_cu = device_cus();
    const int I = d->num_item, A = d->ability_dim;
    pl->msplit = false;
    pl->narrow = false;
    if (A > VIBO_MAX_ABILITY_DIM) {
        // ability_dim 9..16: the wave-per-person kernel's wide instantiation (every row-split / tiled kernel holds 8 dims)
        memset(pl, 0, sizeof(*pl));
        pl->general = true;
        pl->AT = 8;
        pl->D = item_feat_dim(d->irt_model, A);
        pl->lay = partial_layout(A, pl->D, I, d->n_flows);
        pl->total_bytes = 256;            // 8 scalar accumulators
        return 16;
    }
    pl->AT = padded_ability_dim(A);
    pl->D = item_feat_dim(d->irt_model, A);
    pl->DP = prepped_item_width(d->irt_model, pl->AT);
    pl->n_tiles = (d->num_person + kTilePersons - 1) / kTilePersons;
    // wave-per-person kernel: conditional posterior, > 1024 items; planar flows only when the row-split kernel
    // cannot take the launch (ragged / unaligned rows, int64 mask, < 192 items)
    pl->general = d->posterior == VIBO_POSTERIOR_CONDITIONAL || I > 1024;
    const bool split_shape = I >= 4 && I <= 1024 && rows_chunkable(d) && d->mask_dtype != VIBO_MASK_I64;
    if (d->n_flows > 0 && !split_shape) pl->general = true;
    pl->row_ok = false;
    pl->split_ok = false;
    pl->panels = 0;
    pl->cond = false;
    pl->given = false;
    pl->cond_fused = false;
    const bool is_cond = d->posterior == VIBO_POSTERIOR_CONDITIONAL;
    const bool is_given = d->posterior == VIBO_POSTERIOR_GIVEN;
    if (is_given && !(I >= 4 && I <= 32767 && rows_chunkable(d) && d->mask_dtype != VIBO_MASK_I64))
        return fail(-8, "VIBO_POSTERIOR_GIVEN needs the row-split path: 4..32767 items, rows chunkable in 4 cells, no int64 mask");
    if (I >= 4 && I <= 32767 && rows_chunkable(d) && d->mask_dtype != VIBO_MASK_I64 &&
        (is_cond || is_given || (!is_cond && I > 1024))) {
        // panel mode (item counts up to 32767: the whole-row counts are packed as n_correct << 16 | n_observed in an
        // int): one row-split launch per 1024 items (the backward is linear in d LL/d theta, so the panels
        // backpropagate their partial sums independently).  Unconditional posterior: a row-count pass supplies the
        // whole-row counts.  Conditional posterior (any item count): cond_pre_kernel supplies the product-of-experts
        // sums, cond_post_kernel scatters the table gradient (vibo_cond.hip).  The wave-per-person kernel remains
        // the fallback for unaligned rows (decided at launch).
        pl->panels = (I + 1023) / 1024;
        pl->cond = is_cond;
        pl->given = is_given;
        // the conditional posterior's passes on the matrix pipe need the rows as cell codes: the caller's, or the ones the
        // first pass over fp32 rows leaves behind
        // Where they win was measured with hipGraph replays of both forms over persons x items x ability_dim
        // (tools/calibrate_planner.py --cond: VIBO_FLAG_COND_MATRIX against VIBO_FLAG_COND_VALU; profiles/r03_cond_calibration.txt):
        //   rows = cell codes:  5+ dims always (16 x 1 000: 47 vs 67 us -- the VALU passes take two launches each there),
        //                       3-4 dims from 1 024 persons, 2 dims from 4 096, 1 dim from ~16 M cells (16 384 x 1 000: 74 vs 76 us)
        //   rows = fp32:        the VALU pre pass reads the rows AND leaves the codes behind (1M x 1k: 1.22 ms = the 5 B/cell
        //                       stream); a count-and-emit pass in front of the matrix-pipe pre pass costs the same 1.25 ms
        //                       again, so the VALU pre pass stays up to 4 ability dims (one launch: 1M x 1k at 3 / 4 dims
        //                       2.60 -> 2.39 / 2.41 ms) and only the gradient pass moves: 3+ dims always, else from 2 048 persons
        const bool have_codes = d->mask_dtype == VIBO_MASK_CODES || (emit_codes_wanted(d) && d->mask_dtype != VIBO_MASK_I64);
        long long min_persons = (d->flags & VIBO_FLAG_COND_MATRIX) ? 1 : -1;
        if (min_persons < 0) {
            if (d->mask_dtype == VIBO_MASK_CODES) {
                const long long by_cells = 16000000LL / (I > 0 ? I : 1);
                min_persons = A >= 5 ? 1 : A >= 3 ? 1024 : A == 2 ? 4096 : (by_cells > 16384 ? by_cells : 16384);
            } else {
                min_persons = A >= 3 ? 1 : 2048;
            }
        }
        const bool cmat_ok = is_cond && !(d->flags & VIBO_FLAG_COND_VALU) && have_codes && d->num_person >= min_persons;
        pl->cmat_post = cmat_ok && d->want_grad;
        pl->cmat_pre = cmat_ok && (d->mask_dtype == VIBO_MASK_CODES || A >= 5);
        const int at_min = 2;
        if (pl->AT < at_min) pl->AT = at_min;
        pl->DP = prepped_item_width(d->irt_model, pl->AT);
        pl->split_nq = 4;
        pl->split_nblk = num_cu * ((d->want_grad && !codes_three_waves(d, pl->AT)) ? 2 : 3);
        if (pl->split_nblk > (d->num_person + 7) / 8) pl->split_nblk = (d->num_person + 7) / 8;
        pl->cond_nblk = num_cu * 2;      // (4 per CU for one ability dim was tried: the fp32-row variants spill 35-46 registers, 2x slower)
        if (pl->cond_nblk > (d->num_person + 7) / 8) pl->cond_nblk = (d->num_person + 7) / 8;
        pl->cond_post_nblk = pl->cond_nblk;      // (raised below once it is known whether cond_post reads cell codes)
        if (allow_msplit && want_msplit(d)) {
            pl->msplit = true;
            pl->AT = 8;
            pl->DP = prepped_item_width(d->irt_model, 8);
            pl->split_nblk = msplit_blocks(num_cu, I < 1024 ? I : 1024, d->num_person);
            if (pl->panels > 1) {
                // all panels in ONE launch (ElboParams::panel_count): the chip's workgroup slots are shared out over the panels
                int per = num_cu / pl->panels;
                if (per < 1) per = 1;
                if (pl->split_nblk > per) pl->split_nblk = per;
            }
        }
        // Conditional posterior, one panel, ability_dim 1, fp32 rows: the matrix kernel gathers the experts itself while it packs the
        // cells (its XM == 3) and leaves the rows' cell codes behind for the table-gradient pass -- one 5 B/cell stream where
        // cond_pre read 5 + wrote 1 and the matrix kernel read 1 (1M x 1k: 2.28 -> see DESIGN 8.1).  VIBO_FLAG_COND_THREE_PASS /
        // VIBO_FLAG_COND_VALU / VIBO_FLAG_NO_EMIT_CODES keep the three passes.
        pl->cond_fused = is_cond && pl->panels == 1 && A == 1 && pl->msplit && d->n_flows == 0 && d->mask_dtype != VIBO_MASK_CODES &&
                         !(d->flags & (VIBO_FLAG_COND_THREE_PASS | VIBO_FLAG_COND_VALU)) && (!d->want_grad || emit_codes_wanted(d));
        if (pl->cond_fused) pl->cmat_pre = false;
        pl->nblk = 0;
        pl->lds_main = 0;
        pl->lay = partial_layout(A, pl->D, 1024, d->n_flows);
        pl->off_item_prep = 0;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        size_t off = up((size_t)((I + 15) & ~15) * pl->DP * 4);
        pl->off_cnt = off;
        off += up((size_t)d->num_person * 4);
        pl->off_partial = off;
        off += up((size_t)pl->panels * pl->split_nblk * pl->lay.stride * 4);
        pl->cond_rec = 8 * A * 1024;
        pl->off_pre = pl->off_coef = pl->off_cpart = off;
        if (is_cond) {
            pl->off_pre = off;
            off += up((size_t)pl->panels * d->num_person * (2 * A + 1) * 4);
            pl->off_coef = off;
            off += up((size_t)pl->panels * d->num_person * 4 * A * 4);
            pl->off_cpart = off;
            if (pl->cmat_pre || pl->cmat_post) off += up(cond_mfma_scratch_bytes(d->num_person, I, A));
            if (!pl->cmat_post) {
                // cond_post on cell codes (the caller's, or the ones cond_pre leaves behind) has no fp32 row registers: 3 waves per SIMD
                const bool post_codes = d->mask_dtype == VIBO_MASK_CODES ||
                                        (emit_codes_wanted(d) && d->mask_dtype != VIBO_MASK_I64);
                if (post_codes && A <= 2) {
                    pl->cond_post_nblk = num_cu * 3;
                    if (pl->cond_post_nblk > (d->num_person + 7) / 8) pl->cond_post_nblk = (d->num_person + 7) / 8;
                }
                off += up((size_t)pl->panels * pl->cond_post_nblk * pl->cond_rec * 4);
            }
        } else if (is_given) {
            pl->off_pre = off;
            off += up((size_t)d->num_person * (2 * A + 1) * 4);
            pl->off_coef = off;
            off += up((size_t)pl->panels * d->num_person * 4 * A * 4);
        }
        // fp32 rows of the conditional posterior (three passes, five at ability_dim > 4): cond_pre also writes the rows' 1-byte
        // cell codes, the later passes read those: 8 instead of 15 B/term of HBM traffic.  (For the two passes of the
        // unconditional posterior with more than 1024 items the extra write costs more than the cheaper panels win:
        // 100k x 10k 2.32 vs 2.23 ms, so row_count_kernel's code output stays unused there.)
        pl->off_codes = 0;
        pl->codes_stride = ((long long)I + 255) / 256 * 256;      // whole 128-byte lines per wave store (4 B per lane x 64 lanes)
        if (emit_codes_wanted(d) && d->mask_dtype != VIBO_MASK_CODES && d->mask_dtype != VIBO_MASK_I64 && is_cond) {
            pl->off_codes = off;
            off += up((size_t)d->num_person * pl->codes_stride);
        }
        pl->total_bytes = off + 256;
        pl->general = false;
        return 16;
    }
    if (pl->general) {
        pl->nblk = 0;
        pl->lds_main = 0;
        pl->lay = partial_layout(A, pl->D, I, d->n_flows);
        pl->off_item_prep = 0;
        pl->off_partial = 0;
        pl->total_bytes = 256;            // 8 scalar accumulators
        return 16;
    }
    // waves per workgroup (each wave owns <= SB 16-item blocks, see vibo_elbo_kernel.hpp geometry table);
    // the code tile is double-buffered in LDS, 16/waves workgroups share a CU
    const int waves = I <= 144 ? 2 : I <= 304 ? 4 : I <= 512 ? 8 : 16;
    const int stride = code_tile_stride(waves);
    size_t main_b = (size_t)kTilePersons * stride;                    // one fp8 code tile
    const size_t red = (size_t)waves * pl->AT * 65 * 4;               // per-wave dLL/dtheta partials (aliased)
    if (main_b < red) main_b = red;
    if (main_b < (size_t)waves * 32) main_b = (size_t)waves * 32;
    main_b = (main_b + 15) & ~(size_t)15;
    size_t lds = 2 * main_b                                            // double-buffered code tile
                 + 2 * (size_t)(pl->AT + 1) * 65 * 4                   // [theta|valid] share, double-buffered
                 + (size_t)waves * 16 * 20 * 4                         // per-wave G-tile transpose slab
                 + 2 * kTilePersons * 4 + 4 * 2 * pl->AT * 4;          // counts, encoder-table constants
    lds = (lds + 255) & ~(size_t)255;
    const size_t lds_cu = 160 * 1024;
    int per_cu = (int)(lds_cu / lds);
    const int wave_cap = 16 / waves;             // 4 waves per SIMD (launch bound) = 16 per CU
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    int nblk = num_cu * per_cu;
    if (nblk > pl->n_tiles) nblk = pl->n_tiles;
    pl->nblk = nblk;
    // wave-per-row kernel (A <= 2, 1PL/2PL, 192 <= I <= 1024): 4 workgroups of 4 waves per CU
    pl->row_ok = A <= 2 && d->irt_model <= 2 && I >= 192 && I <= 1024 && (I % 4 == 0) && d->n_flows == 0;
    pl->row_nblk = num_cu * 2;          // 16 items x (params + grads) per lane: 2 workgroups (8 waves) per CU
    if (pl->row_nblk > (d->num_person + 3) / 4) pl->row_nblk = (d->num_person + 3) / 4;
    // row-split kernel (192 <= I <= 1024, u8 / no mask): nq waves share a row, 8 waves per CU.
    // Preferred over the wave-per-row kernel (1.03 vs 1.10 ms at A = 1, 1.03 vs 1.49 ms at A = 2 on 1M x 1k),
    // which stays for int64 masks.
    pl->split_ok = split_shape;
    pl->split_nq = (I + 255) / 256;
    if (pl->split_ok) {       // the row-split kernel's narrowest template is 2 wide
        const int at_min = 2;
        if (pl->AT < at_min) {
            pl->AT = at_min;
            pl->DP = prepped_item_width(d->irt_model, at_min);
        }
    }
    pl->split_nblk = num_cu * (((d->want_grad && !codes_three_waves(d, pl->AT)) ? 8 : 12) / pl->split_nq);   // forward-only fits 3 waves per SIMD
    if (pl->split_nblk > (d->num_person + 7) / 8) pl->split_nblk = (d->num_person + 7) / 8;
    if (pl->split_ok && allow_msplit && want_msplit(d)) {
        pl->msplit = true;
        pl->AT = 8;
        pl->DP = prepped_item_width(d->irt_model, 8);
        pl->split_nblk = msplit_blocks(num_cu, I, d->num_person);
    }
    if (pl->split_ok && allow_msplit && !pl->msplit && want_narrow(d)) {
        pl->narrow = true;
        pl->split_nblk = narrow_blocks(num_cu, d);
    }
    pl->lds_main = (int)main_b;
    pl->geom.waves = waves;
    pl->geom.grid = nblk;
    pl->geom.lds_bytes = lds;
    pl->lay = partial_layout(A, pl->D, I, d->n_flows);
    pl->off_item_prep = 0;
    size_t prep_bytes = ((size_t)((I + 15) & ~15) * pl->DP * 4 + 255) & ~(size_t)255;
    pl->off_partial = prep_bytes;
    int max_blk = nblk;
    if (pl->row_ok && pl->row_nblk > max_blk) max_blk = pl->row_nblk;
    if (pl->split_ok && pl->split_nblk > max_blk) max_blk = pl->split_nblk;
    pl->total_bytes = prep_bytes + (size_t)max_blk * pl->lay.stride * 4 + 256;
    return stride;
}

// ---------------------------------------------------------------------------
// item prep: [I][D] item sample -> [I][DP] rows the fused kernel reads with scalar loads
// ---------------------------------------------------------------------------
__global__ void item_prep_kernel(const float* __restrict__ item, float* __restrict__ prep, int I, int A, int AT,
                                 int D, int DP, int irt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int I16 = (I + 15) & ~15;
    if (i >= I16) return;
    float* dst = prep + (size_t)i * DP;
    if (i >= I) {            // zero rows pad the item axis to a multiple of 16
        for (int a = 0; a < DP; ++a) dst[a] = 0.f;
        return;
    }
    const float* src = item + (size_t)i * D;
    // logits are carried in log2 units (x log2 e) so the kernel's exp2/log2 need no extra multiply
    for (int a = 0; a < DP; ++a) dst[a] = 0.f;
    if (irt == 1) {          // logit = sum_a theta_a + b   (models.py:731)
        for (int a = 0; a < A; ++a) dst[a] = kLog2e;
        dst[AT] = src[0] * kLog2e;
        return;
    }
    for (int a = 0; a < A; ++a) dst[a] = -src[a] * kLog2e;   // logit = -a.theta + b   (models.py:744,759)
    dst[AT] = src[A] * kLog2e;
    if (irt == 3) {
        const float g = 1.0f / (1.0f + expf(-src[A + 1]));   // guess = sigmoid(guess logit) (models.py:758)
        dst[AT + 1] = g;
        dst[AT + 2] = 1.0f - g;
    }
}

// ---------------------------------------------------------------------------
// panel mode: packed counts (n_correct << 16 | n_observed) of every person row, one wave per row
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_count_kernel(const float* __restrict__ response, const void* __restrict__ mask,
                                                        const int64_t* __restrict__ row_index, int* __restrict__ cnt,
                                                        long long resp_stride, long long mask_stride, int B, int I,
                                                        int mask_dtype, uint8_t* __restrict__ codes_out = nullptr,
                                                        long long codes_stride = 0) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * 4;
    const int n4 = (I + 3) >> 2;
    const bool cell_codes = mask_dtype == VIBO_MASK_CODES;
    for (long long row = wave_id; row < B; row += n_waves) {
        const long long src = row_index ? row_index[row] : row;
        const float4* rp = reinterpret_cast<const float4*>(response + src * resp_stride);
        const uint32_t* mp = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(mask) + src * mask_stride);
        int packed = 0;
        for (int c0 = lane; c0 < n4; c0 += 256) {          // 4 chunks per lane in flight
            float4 x[4];
            uint32_t m[4], keep[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 64 * u;
                x[u] = float4{0.f, 0.f, 0.f, 0.f};
                m[u] = 0u;
                keep[u] = 0u;
                if (c < n4) {
                    if (!cell_codes) x[u] = nt_load4(rp + c);       // (streamed once: see nt_load4)
                    m[u] = (mask_dtype == 0 || cell_codes) ? mp[c] : 0x01010101u;
                    keep[u] = ((I & 3) && c == (I >> 2)) ? (1u << (8 * (I & 3))) - 1u : 0xFFFFFFFFu;      // padded tail of the row
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (cell_codes) (void)pack_cell_codes4(m[u], keep[u], packed);
                else (void)pack_codes4(x[u], m[u] & keep[u], packed);
                // fp32 rows, more passes to come: leave the row behind as 1-byte cell codes (minibatch order)
                if (!cell_codes && codes_out && c0 + 64 * u < n4)
                    reinterpret_cast<uint32_t*>(codes_out + row * codes_stride)[c0 + 64 * u] = cell_codes4(x[u], m[u] & keep[u]);
            }
        }
        const int tot = lane63(wave_sum63(packed));
        if (lane == 0) cnt[row] = tot;
    }
}

// ---------------------------------------------------------------------------
// VIBO_POSTERIOR_GIVEN: the caller's per-person (mu | logvar) enters the row-split kernel through the hooks of the
// conditional pipeline: precision lam = exp(-logvar), s = mu lam, nobs = I (no prior experts are added); the kernel's
// per-person coefficients P1 = g_mu / lam, P2 = -(g_mu mu + g_lv) / lam come back as d / d (mu, logvar)
// ---------------------------------------------------------------------------
// conditional posterior, more than one 1024-item panel: the panels' row statistics summed once (fixed order) into panel 0's
// block, so the matrix kernel's per-person forward reads 3 values instead of 3 per panel inside its barrier phase
__global__ __launch_bounds__(256) void panel_sum_kernel(float* __restrict__ pre, long long n, int panels) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float t = pre[e];
    for (int pn = 1; pn < panels; ++pn) t += pre[(size_t)pn * n + e];
    pre[e] = t;
}

__global__ __launch_bounds__(256) void given_pre_kernel(const float* __restrict__ post, float* __restrict__ pre, long long B, int A,
                                                        int I) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= B * (A + 1)) return;
    const long long row = e / (A + 1);
    const int a = (int)(e % (A + 1));
    float* st = pre + row * (2 * A + 1);
    if (a == A) { st[2 * A] = (float)I; return; }
    const float mu = post[row * 2 * A + a], lam = expf(-post[row * 2 * A + A + a]);
    st[a] = lam;
    st[A + a] = mu * lam;
}
__global__ __launch_bounds__(256) void given_post_kernel(const float* __restrict__ post, const float* __restrict__ coef, int panels,
                                                         float* __restrict__ grad, long long B, int A) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= B * A) return;
    const long long row = e / A;
    const int a = (int)(e % A);
    const float mu = post[row * 2 * A + a], lam = expf(-post[row * 2 * A + A + a]);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float p1 = 0.f, p2 = 0.f;
        for (int pn = 0; pn < panels; ++pn) {
            const float* pc = coef + ((size_t)pn * B + row) * 4 * A;
            p1 += pc[(st * 2 + 0) * A + a];
            p2 += pc[(st * 2 + 1) * A + a];
        }
        const float gmu = p1 * lam;
        grad[((size_t)st * B + row) * 2 * A + a] = gmu;
        grad[((size_t)st * B + row) * 2 * A + A + a] = -p2 * lam - gmu * mu;
    }
}

// whole-row counts for rows the vector kernel cannot read (unaligned / not chunkable / int64 mask): wave per row
__global__ __launch_bounds__(256) void row_count_scalar_kernel(const float* __restrict__ response, const void* __restrict__ mask,
                                                               const int64_t* __restrict__ row_index, int* __restrict__ cnt,
                                                               long long resp_stride, long long mask_stride, int B, int I,
                                                               int mask_dtype) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const long long src = row_index ? row_index[row] : row;
    int packed = 0;
    for (int i = lane; i < I; i += 64) {
        bool k = true, one;
        if (mask_dtype == VIBO_MASK_CODES) {
            const uint8_t c = static_cast<const uint8_t*>(mask)[src * mask_stride + i];
            k = c != 2; one = c == 1;
        } else {
            if (mask_dtype == VIBO_MASK_U8) k = static_cast<const uint8_t*>(mask)[src * mask_stride + i] != 0;
            else if (mask_dtype == VIBO_MASK_I64) k = static_cast<const int64_t*>(mask)[src * mask_stride + i] != 0;
            one = response[src * resp_stride + i] == 1.0f;
        }
        if (k) packed += 1 + (one ? (1 << 16) : 0);
    }
    const int tot = lane63(wave_sum63(packed));
    if (lane == 0) cnt[row] = tot;
}

// ---------------------------------------------------------------------------
// Format P: response (fp32) + mask (u8 / int64 / none) -> 1-byte cell codes, rows padded with "missing" up to the
// code stride (datasets.py:928-940 stores responses as fp32 with -1 for missing and a separate mask)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_codes_kernel(const float* __restrict__ response, const void* __restrict__ mask,
                                                         uint8_t* __restrict__ codes, long long resp_stride, long long mask_stride,
                                                         long long code_stride, long long B, int I, int mask_dtype) {
    const long long n = B * code_stride;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long row = e / code_stride;
        const int i = (int)(e - row * code_stride);
        uint8_t c = 2;
        if (i < I) {
            bool k = true;
            if (mask_dtype == VIBO_MASK_U8) k = static_cast<const uint8_t*>(mask)[row * mask_stride + i] != 0;
            else if (mask_dtype == VIBO_MASK_I64) k = static_cast<const int64_t*>(mask)[row * mask_stride + i] != 0;
            if (k) c = response[row * resp_stride + i] == 1.0f ? 1 : 0;
        }
        codes[e] = c;
    }
}

// the same for 4-cell chunks of aligned rows (thread = one chunk: float4 + mask word in, one code word out)
__global__ __launch_bounds__(256) void pack_codes4_kernel(const float* __restrict__ response, const void* __restrict__ mask,
                                                          uint32_t* __restrict__ codes, long long resp_stride, long long mask_stride,
                                                          long long chunks_per_row, long long B, int I, int mask_dtype) {
    const long long n = B * chunks_per_row;
    const int n4 = (I + 3) >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long row = e / chunks_per_row;
        const int c = (int)(e - row * chunks_per_row);
        uint32_t w = kAllMissing4;
        if (c < n4) {
            const float4 x = reinterpret_cast<const float4*>(response + row * resp_stride)[c];
            uint32_t m = mask_dtype == VIBO_MASK_U8
                             ? reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(mask) + row * mask_stride)[c]
                             : 0x01010101u;
            m = ((m | (m >> 1) | (m >> 2) | (m >> 3) | (m >> 4) | (m >> 5) | (m >> 6) | (m >> 7)) & 0x01010101u);   // any bit -> 1
            if ((I & 3) && c == (I >> 2)) m &= (1u << (8 * (I & 3))) - 1u;
            const uint32_t one = (x.x == 1.0f ? 1u : 0u) | (x.y == 1.0f ? 1u << 8 : 0u) | (x.z == 1.0f ? 1u << 16 : 0u) |
                                 (x.w == 1.0f ? 1u << 24 : 0u);
            w = (one & m) | ((m ^ 0x01010101u) << 1);          // observed: 0 / 1; not observed (or padding): 2
        }
        codes[e] = w;
    }
}

// ---------------------------------------------------------------------------
// finalize: fixed-order sum of the per-block partial records (fp64 accumulate)
// ---------------------------------------------------------------------------
// 1024 threads = OUT outputs x (1024 / OUT) slices of the block list; OUT = 64 normally, 16 when there are many small
// records (one-wave workgroups of the row-split kernel: up to 8 per CU)
template <int OUT>
__global__ __launch_bounds__(1024) void finalize_kernel(const FinalizeParams f) {
    constexpr int SLICES = 1024 / OUT;
    __shared__ __attribute__((aligned(16))) double part[SLICES][OUT];
    if ((int)blockIdx.x >= f.n_fin) {      // the conditional posterior's table gradients (vibo_cond_finalize.hpp)
        cond_fin_tail_body(f.tail, (int)blockIdx.x - f.n_fin, &part[0][0]);
        return;
    }
    // element e of the logical output vector: [0,8) scalars | table grads | flow grads | item grads
    const int n_tab = 8 * f.A;
    const int n_flow = 2 * f.n_flows * (2 * f.A + 1);
    const int n_item = f.I * f.D;
    const int n_out = 8 + (f.want_grad ? n_tab + n_flow + n_item : 0);
    const int lane = threadIdx.x % OUT, slice = threadIdx.x / OUT;
    const int e = blockIdx.x * OUT + lane;
    double acc = 0.0;
    if (e < n_out) {
        int src, b0 = 0, b1 = f.nblk;
        if (e < 8 + n_tab + n_flow) {
            src = e;   // same offsets in the partial record (off_table = 8, off_flow = 8 + 8A)
        } else {
            const int k = e - (8 + n_tab + n_flow);
            const int dd = k / f.I, i = k % f.I;          // consecutive lanes = consecutive items: coalesced record reads
            const int panel = i / f.panel_items;          // panel mode: only this panel's blocks hold item i
            src = f.lay.off_item + dd * f.lay.i_pad + (i - panel * f.panel_items);
            b0 = panel * f.bpp;
            b1 = b0 + f.bpp;
        }
        // fixed order: slice s sums blocks s, s+SLICES, ... in fp64, then the slices are summed in order
        acc = record_slice_sum<SLICES>(f.partial, (size_t)f.lay.stride, src, b0, b1, slice);
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && e < n_out) {
        double t = 0.0;
#pragma unroll
        for (int s = 0; s < SLICES; ++s) t += part[s][lane];
        if (e < 8) {
            // partial scalars: 0 ll, 1 kl, 2 logq0, 3 logp, 4 ladj, 5 nobs
            part[0][lane] = t;
        } else if (e < 8 + n_tab) {
            if (f.grad_table) f.grad_table[e - 8] = (float)t;      // null: conditional posterior (cond_finalize_kernel)
        } else if (e < 8 + n_tab + n_flow) {
            f.grad_flow[e - 8 - n_tab] = (float)t;
        } else {
            const int k = e - 8 - n_tab - n_flow;
            f.grad_item[(size_t)(k % f.I) * f.D + k / f.I] = (float)t;
        }
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const double ll = part[0][0], kl = part[0][1], logq0 = part[0][2], logp = part[0][3], ladj = part[0][4];
            f.out_scalars[VIBO_S_LL] = (float)ll;
            f.out_scalars[VIBO_S_REG] = (float)(f.reg_mode == VIBO_REG_KL ? kl : (logq0 - ladj - logp));
            f.out_scalars[VIBO_S_KL] = (float)kl;
            f.out_scalars[VIBO_S_LOGQ0] = (float)logq0;
            f.out_scalars[VIBO_S_LOGP] = (float)logp;
            f.out_scalars[VIBO_S_LADJ] = (float)ladj;
            f.out_scalars[VIBO_S_NOBS] = (float)part[0][5];
            f.out_scalars[VIBO_S_RESERVED] = 0.f;
        }
    }
}

// multi-sample forward: out_scalars[s][8] from the per-block records (8 scalars per sample at record[8 s ..])
__global__ __launch_bounds__(1024) void multi_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out_scalars,
                                                              int nblk, int stride, int n_samples, int reg_mode) {
    __shared__ double part[32][32];
    const int e = threadIdx.x & 31, slice = threadIdx.x >> 5;      // e = 8 s + k
    double acc = 0.0;
    if (e < 8 * n_samples)
        for (int b = slice; b < nblk; b += 32) acc += (double)partial[(size_t)b * stride + e];
    part[slice][e] = acc;
    __syncthreads();
    if (slice == 0) {
        double t = 0.0;
        for (int s = 0; s < 32; ++s) t += part[s][e];
        part[0][e] = t;
    }
    __syncthreads();
    if (threadIdx.x < n_samples) {
        const int s = threadIdx.x;
        const double ll = part[0][8 * s + 0], kl = part[0][8 * s + 1], logq0 = part[0][8 * s + 2];
        const double logp = part[0][8 * s + 3], ladj = part[0][8 * s + 4];
        float* o = out_scalars + 8 * s;
        o[VIBO_S_LL] = (float)ll;
        o[VIBO_S_REG] = (float)(reg_mode == VIBO_REG_KL ? kl : (logq0 - ladj - logp));
        o[VIBO_S_KL] = (float)kl;
        o[VIBO_S_LOGQ0] = (float)logq0;
        o[VIBO_S_LOGP] = (float)logp;
        o[VIBO_S_LADJ] = (float)ladj;
        o[VIBO_S_NOBS] = (float)part[0][8 * s + 5];
        o[VIBO_S_RESERVED] = 0.f;
    }
}

// ---------------------------------------------------------------------------
// forward-only encode: one wave per person (models.py:356-371 under no_grad)
// ---------------------------------------------------------------------------
struct EncodeParams {
    const float* response;
    const void* mask;
    const int64_t* row_index;
    const float* table;
    float* ability_mu;
    float* ability_logvar;
    long long resp_stride, mask_stride;
    int B, I, A, mask_dtype, missing_mode, conditional;
};

__global__ __launch_bounds__(256) void encode_kernel(const EncodeParams p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B) return;
    const long long src = p.row_index ? p.row_index[row] : row;
    const float* rp = p.response + src * p.resp_stride;
    const int A = p.A, I = p.I;
    float lam[VIBO_MAX_ABILITY_DIM_WIDE], smu[VIBO_MAX_ABILITY_DIM_WIDE];
#pragma unroll
    for (int a = 0; a < VIBO_MAX_ABILITY_DIM_WIDE; ++a) lam[a] = smu[a] = 0.f;
    const float tau_prior = 1.0f / (1.0f + kPoeEps);
    for (int i = lane; i < I; i += 64) {
        bool k;
        if (p.mask_dtype == VIBO_MASK_U8) k = static_cast<const uint8_t*>(p.mask)[src * p.mask_stride + i] != 0;
        else if (p.mask_dtype == VIBO_MASK_I64) k = static_cast<const int64_t*>(p.mask)[src * p.mask_stride + i] != 0;
        else k = true;
        const int c = (rp[i] == 1.0f) ? 1 : 0;
        const float* te = p.conditional ? p.table + ((size_t)c * I + i) * 2 * A : p.table + (size_t)c * 2 * A;
#pragma unroll
        for (int a = 0; a < VIBO_MAX_ABILITY_DIM_WIDE; ++a) {
            if (a < A) {
                if (k) {
                    const float tau = 1.0f / (__expf(te[A + a]) + kPoeEps);
                    lam[a] += tau;
                    smu[a] = fmaf(te[a], tau, smu[a]);
                } else if (p.missing_mode == VIBO_MISSING_PRIOR) {
                    lam[a] += tau_prior;
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < VIBO_MAX_ABILITY_DIM_WIDE; ++a) {
        if (a < A) {
            const float L = wave_total(lam[a]);
            const float S = wave_total(smu[a]);
            if (lane == 0) {
                p.ability_mu[row * A + a] = S / L;
                p.ability_logvar[row * A + a] = logf(1.0f / L);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// decode: response_mu[B][I] (models.py:729-766)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ ability, const float* __restrict__ item,
                                                     float* __restrict__ out, int B, int I, int A, int D, int irt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = blockIdx.y;
    if (i >= I) return;
    const float* th = ability + b * A;
    const float* it = item + (size_t)i * D;
    float logit;
    if (irt == 1) {
        logit = it[0];
        for (int a = 0; a < A; ++a) logit += th[a];
    } else {
        logit = it[A];
        for (int a = 0; a < A; ++a) logit = fmaf(-it[a], th[a], logit);
    }
    float pr = 1.0f / (1.0f + expf(-logit));
    if (irt == 3) {
        const float g = 1.0f / (1.0f + expf(-it[A + 1]));
        pr = g + (1.0f - g) * pr;
    }
    out[b * I + i] = pr;
}

// forward-only posterior from whole-row statistics: thread = (person, ability dim).  stats = packed counts of
// row_count_kernel (unconditional: the experts are the two table rows) or the per-panel sums of cond_pre_kernel.
__global__ __launch_bounds__(256) void encode_finish_kernel(const int* __restrict__ cnt, const float* __restrict__ pre, int panels,
                                                            const float* __restrict__ table, float* __restrict__ ability_mu,
                                                            float* __restrict__ ability_logvar, long long B, int I, int A,
                                                            int missing_mode) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= B * A) return;
    const long long row = e / A;
    const int a = (int)(e % A);
    float lam, smu, nobs;
    if (cnt) {
        const int c = cnt[row];
        const float n1 = (float)(c >> 16);
        nobs = (float)(c & 0xffff);
        const float n0 = nobs - n1;
        const float tau0 = 1.0f / (expf(table[A + a]) + kPoeEps), tau1 = 1.0f / (expf(table[2 * A + A + a]) + kPoeEps);
        lam = n0 * tau0 + n1 * tau1;
        smu = n0 * table[a] * tau0 + n1 * table[2 * A + a] * tau1;
    } else {
        lam = smu = nobs = 0.f;
        for (int pn = 0; pn < panels; ++pn) {
            const float* st = pre + ((size_t)pn * B + row) * (2 * A + 1);
            lam += st[a]; smu += st[A + a]; nobs += st[2 * A];
        }
    }
    if (missing_mode == VIBO_MISSING_PRIOR) lam += ((float)I - nobs) * (1.0f / (1.0f + kPoeEps));
    ability_mu[e] = smu / lam;
    ability_logvar[e] = logf(1.0f / lam);
}

// conditional posterior on cell codes, 4 096 persons or more: the experts' sums on the matrix pipe (launch_cond_pre_mfma), as in the
// ELBO call
static bool encode_on_matrix_pipe(const vibo_desc* d) {
    return d->posterior == VIBO_POSTERIOR_CONDITIONAL && d->mask_dtype == VIBO_MASK_CODES && !(d->flags & VIBO_FLAG_COND_VALU) &&
           (d->num_person >= 4096 || d->ability_dim >= 5 || (d->flags & VIBO_FLAG_COND_MATRIX));
}
// scratch the fast encode path needs (0: not applicable -> wave-per-person encode_kernel)
static size_t encode_scratch_bytes(const vibo_desc* d) {
    const int I = d->num_item, A = d->ability_dim;
    if (A > VIBO_MAX_ABILITY_DIM) return 0;      // (wave-per-person encode kernel)
    if (I < 4 || I > 32767 || !rows_chunkable(d) || d->mask_dtype == VIBO_MASK_I64) return 0;
    if (d->posterior == VIBO_POSTERIOR_CONDITIONAL) {
        size_t pre = ((size_t)((I + 1023) / 1024) * d->num_person * (2 * A + 1) * 4 + 255) & ~(size_t)255;
        if (encode_on_matrix_pipe(d)) pre += cond_mfma_scratch_bytes(d->num_person, I, A);      // (only its table image is used)
        return pre + 256;
    }
    return (size_t)d->num_person * 4 + 256;
}

// posterior-predictive mean: thread = one item x 8 persons; per sample the item row is loaded once and reused for the
// 8 persons (ability rows are wave-uniform scalar loads)
__global__ __launch_bounds__(256) void decode_mean_kernel_strided(const float* __restrict__ ability, const float* __restrict__ item,
                                                                  float* __restrict__ out, int S, int B, int B_total, int I, int A,
                                                                  int D, int irt) {
    constexpr int RB = 8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long b0 = (long long)blockIdx.y * RB;
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.f;
    const bool ok = i < I;
    for (int s = 0; s < S; ++s) {
        const float* it = item + ((size_t)s * I + (ok ? i : 0)) * D;
        float a[VIBO_MAX_ABILITY_DIM_WIDE];
#pragma unroll
        for (int k = 0; k < VIBO_MAX_ABILITY_DIM_WIDE; ++k) a[k] = (irt != 1 && k < A) ? it[k] : 0.f;
        const float bb = irt == 1 ? it[0] : it[A];
        float g = 0.f;
        if (irt == 3) g = 1.0f / (1.0f + expf(-it[A + 1]));
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const long long b = b0 + r;
            if (b >= B) break;
            const float* th = ability + ((size_t)s * B_total + b) * A;      // sample stride = all persons
            float logit = bb;
            if (irt == 1) {
                for (int k = 0; k < A; ++k) logit += th[k];
            } else {
#pragma unroll
                for (int k = 0; k < VIBO_MAX_ABILITY_DIM_WIDE; ++k)
                    if (k < A) logit = fmaf(-a[k], th[k], logit);
            }
            float pr = 1.0f / (1.0f + expf(-logit));
            if (irt == 3) pr = g + (1.0f - g) * pr;
            acc[r] += pr;
        }
    }
    if (ok) {
        const float inv = 1.0f / (float)S;
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (b0 + r < B) out[(b0 + r) * I + i] = acc[r] * inv;
    }
}

}  // namespace vibo

using namespace vibo;

// xor16_add / xor32_add (v_permlane16_swap / v_permlane32_swap through inline asm, vibo_device.hpp) next to the __shfl_xor form
// they replace: out[0][lane] | out[1][lane] = the swap forms, out[2] | out[3] = the shuffle forms (tests/test_gpu_parity.py)
__global__ void lane_swap_selftest_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const float v = in[lane];
    out[lane] = xor16_add(v);
    out[64 + lane] = xor32_add(v);
    out[128 + lane] = v + __shfl_xor(v, 16);
    out[192 + lane] = v + __shfl_xor(v, 32);
    // chained, as the kernels use them (the second swap reads the first one's fresh result)
    out[256 + lane] = xor32_add(xor16_add(v));
    out[320 + lane] = [&] { const float t = v + __shfl_xor(v, 16); return t + __shfl_xor(t, 32); }();
}

extern "C" {

int vibo_version(void) { return VIBO_ABI_VERSION; }

int vibo_plan_cond_passes(const vibo_desc* d) {
    int rc = check_desc(d);
    if (rc) return rc;
    Plan pl;
    rc = make_plan(d, &pl);
    if (rc < 0) return rc;
    if (pl.general || !pl.cond) return 0;
    return (pl.cmat_pre ? 1 : 0) | (pl.cmat_post ? 2 : 0) | (pl.cond_fused ? 4 : 0);
}

int vibo_plan_kernel(const vibo_desc* d) {
    int rc = check_desc(d);
    if (rc) return rc;
    Plan pl;
    rc = make_plan(d, &pl);
    if (rc < 0) return rc;
    if (pl.general) return VIBO_KERNEL_GENERAL;
    if (pl.panels > 0 || pl.split_ok) return pl.msplit ? VIBO_KERNEL_MATRIX : pl.narrow ? VIBO_KERNEL_NARROW : VIBO_KERNEL_VALU;
    if (pl.row_ok && d->num_item % 4 == 0 && pl.AT == d->ability_dim) return VIBO_KERNEL_ROW;
    return VIBO_KERNEL_TILED;
}

const char* vibo_last_error_string(void) { return g_err; }

int vibo_selftest_lane_swaps(const float* in, float* out, void* stream) {
    if (!in || !out) return fail(-5, "null required pointer");
    hipLaunchKernelGGL(lane_swap_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "lane swap selftest launch");
    return 0;
}

int vibo_set_insitu_timer(uint64_t* block) {
    if ((uintptr_t)block & 7) return fail(-5, "vibo_set_insitu_timer: the block must be 8-byte aligned");
    g_insitu = reinterpret_cast<unsigned long long*>(block);
    return 0;
}
int vibo_insitu_timer_reset(uint64_t* block, void* stream) {
    if (!block || ((uintptr_t)block & 7)) return fail(-5, "vibo_insitu_timer_reset: null / unaligned block");
    // words 0 (earliest entry) and 5 (shortest launch) start at all-ones, the rest at zero: two byte-fills, no kernel
    hipError_t e = hipMemsetAsync(block, 0, 8 * sizeof(uint64_t), (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemsetAsync(block, 0xff, sizeof(uint64_t), (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemsetAsync(block + 5, 0xff, sizeof(uint64_t), (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "insitu timer reset");
    return 0;
}

size_t vibo_workspace_bytes(const vibo_desc* d) {
    if (check_desc(d) != 0) return 0;
    Plan pl;
    if (make_plan(d, &pl) < 0) return 0;
    const size_t enc = encode_scratch_bytes(d);
    return pl.total_bytes > enc ? pl.total_bytes : enc;
}

}  // extern "C"

// the folded train step (vibo_elbo_fwd_bwd_step + vibo_train_epilogue_fused) covers single-launch row-split calls of the plain model
static bool step_plan_ok(const vibo_desc* d, const Plan& pl) {
    return d->posterior == VIBO_POSTERIOR_UNCONDITIONAL && d->n_flows == 0 && d->reg_mode == VIBO_REG_KL && d->want_grad &&
           !pl.general && pl.panels == 0 && pl.split_ok;
}

// counts of the call's rows from the caller's per-source-row counts (vibo_elbo_fwd_bwd_counts with a row_index)
__global__ __launch_bounds__(256) void gather_counts_kernel(const int32_t* __restrict__ all, const int64_t* __restrict__ row_index,
                                                            int* __restrict__ out, int B) {
    const int k = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (k < B) out[k] = all[row_index[k]];
}

static int elbo_fwd_bwd_impl(const vibo_desc* d, int32_t* step_count, int skip_finalize, const float* response, const void* mask,
                             const int64_t* row_index, const float* table, const float* item, const float* eps, const float* flow,
                             float* out_scalars, float* ability_mu, float* ability_logvar, float* ability,
                             float* ability_k, float* ability_ladj, float* grad_table, float* grad_item,
                             float* grad_flow, void* workspace, size_t workspace_bytes, void* stream, const int32_t* row_counts = nullptr) {
    const int num_cu = device_cus();
    int rc = check_desc(d);
    if (rc) return rc;
    if ((!response && d->mask_dtype != VIBO_MASK_CODES) || !table || !item || !eps || !out_scalars || !ability_mu || !ability_logvar || !ability)
        return fail(-5, "null required pointer");
    if ((d->mask_dtype == VIBO_MASK_NONE) != (mask == nullptr)) return fail(-5, "mask pointer / mask_dtype mismatch");
    if (d->want_grad && (!grad_table || !grad_item)) return fail(-5, "want_grad needs grad_table and grad_item");
    if (d->n_flows > 0 && (!flow || !ability_k || !ability_ladj)) return fail(-5, "flows need flow, ability_k, ability_ladj");
    if (d->n_flows > 0 && d->want_grad && !grad_flow) return fail(-5, "want_grad with flows needs grad_flow");
    Plan pl;
    const int stride = make_plan(d, &pl);
    if (stride < 0) return stride;
    if (!workspace || workspace_bytes < pl.total_bytes)
        return fail(-7, "workspace too small: %zu < %zu", workspace_bytes, pl.total_bytes);
    if ((uintptr_t)workspace & 255) return fail(-7, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int I = d->num_item, A = d->ability_dim;
    CondFinTail tail;                  // the conditional posterior's last stage, launched with the ELBO finalize
    memset(&tail, 0, sizeof(tail));
    // 16-byte row loads need aligned rows
    const bool vec = rows_vec_ok(d, response, mask);
    bool codes = d->mask_dtype == VIBO_MASK_CODES;      // (panel mode on fp32 rows: true from the second pass on, see off_codes)
    if (codes && !(vec && !pl.general && (pl.panels > 0 || pl.split_ok))) return codes_unsupported();
    if (pl.given && !vec) return fail(-8, "VIBO_POSTERIOR_GIVEN: rows must be aligned for 4-cell chunks (see vibo_amd.ops.pad_rows)");
    if ((step_count || skip_finalize) && !(step_plan_ok(d, pl) && vec))
        return fail(-8, "vibo_elbo_fwd_bwd_step: single-launch row-split calls of the plain model only (unconditional posterior, no "
                        "flows, KL regulariser, gradients, 4..1024 items, aligned rows): use vibo_train_prologue + vibo_elbo_fwd_bwd");
    if (pl.general || (d->n_flows > 0 && !(pl.split_ok && vec) && pl.panels == 0) || (pl.panels > 0 && !vec)) {
        const size_t n_table = (size_t)(d->posterior == VIBO_POSTERIOR_CONDITIONAL ? 2 * I * 2 * A : 2 * 2 * A);
        const size_t n_flow = (size_t)d->n_flows * (2 * A + 1);
        hipError_t ge = hipMemsetAsync(workspace, 0, 256, s);
        if (ge == hipSuccess && d->want_grad) {
            ge = hipMemsetAsync(grad_table, 0, 2 * n_table * sizeof(float), s);
            if (ge == hipSuccess) ge = hipMemsetAsync(grad_item, 0, (size_t)I * pl.D * sizeof(float), s);
            if (ge == hipSuccess && n_flow) ge = hipMemsetAsync(grad_flow, 0, 2 * n_flow * sizeof(float), s);
        }
        if (ge != hipSuccess) return hip_fail(ge, "memset");
        GeneralParams g;
        memset(&g, 0, sizeof(g));
        g.response = response; g.mask = mask; g.row_index = row_index; g.table = table; g.item = item; g.eps = eps;
        g.flow = flow; g.ability_mu = ability_mu; g.ability_logvar = ability_logvar; g.ability = ability;
        g.ability_k = ability_k; g.ability_ladj = ability_ladj;
        g.grad_table = grad_table; g.grad_item = grad_item; g.grad_flow = grad_flow;
        g.acc_scalars = static_cast<float*>(workspace); g.out_scalars = out_scalars;
        g.resp_stride = d->response_row_stride; g.mask_stride = d->mask_row_stride;
        g.B = d->num_person; g.I = I; g.A = A; g.D = pl.D; g.irt = d->irt_model;
        g.conditional = d->posterior == VIBO_POSTERIOR_CONDITIONAL; g.missing_mode = d->missing_mode;
        g.mask_dtype = d->mask_dtype; g.reg_mode = d->reg_mode; g.n_flows = d->n_flows; g.want_grad = d->want_grad;
        ge = launch_elbo_general(g, num_cu, s);
        if (ge != hipSuccess) return hip_fail(ge, "general elbo kernel launch");
        return 0;
    }

    float* item_prep = reinterpret_cast<float*>(static_cast<char*>(workspace) + pl.off_item_prep);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + pl.off_partial);
    hipError_t e = hipSuccess;
    const bool row_split = pl.panels > 0 || (pl.split_ok && vec);
    if (!row_split) {            // (the row-split kernels read the item sample themselves)
        hipLaunchKernelGGL(item_prep_kernel, dim3((I + 15 + 255) / 256), dim3(256), 0, s, item, item_prep, I, A, pl.AT, pl.D,
                           pl.DP, d->irt_model);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "item_prep launch");
    }

    ElboParams p;
    memset(&p, 0, sizeof(p));
    p.response = response; p.mask = mask; p.row_index = row_index;
    p.table = table; p.item_prep = item_prep; p.item_raw = item; p.eps = eps;
    p.ability_mu = ability_mu; p.ability_logvar = ability_logvar; p.ability = ability;
    p.partial = partial;
    p.resp_stride = d->response_row_stride; p.mask_stride = d->mask_row_stride;
    p.B = d->num_person; p.I = I; p.A = A; p.D = pl.D; p.DP = pl.DP;
    p.n_tiles = pl.n_tiles; p.lds_stride = stride; p.lds_main = pl.lds_main;
    p.mask_dtype = d->mask_dtype; p.missing_mode = d->missing_mode; p.reg_mode = d->reg_mode;
    p.flow = flow; p.ability_k = ability_k; p.ability_ladj = ability_ladj; p.n_flows = d->n_flows;
    p.lay = pl.lay;
    p.vec_ok = (vec && I % 4 == 0) ? 1 : 0;      // the tiled / wave-per-row kernels' vector loads assume whole chunks
    p.row_cnt = nullptr; p.item0 = 0; p.I_total = I; p.primary = 1;
    p.step_tick = step_count;
    p.insitu = g_insitu;

    const bool grad = d->want_grad != 0;
    int nblk_used = pl.nblk;
    int panel_items = 1 << 30, bpp = 0;
    if (pl.panels > 0) {
        char* wsb = static_cast<char*>(workspace);
        float* pre = reinterpret_cast<float*>(wsb + pl.off_pre);
        float* coef = reinterpret_cast<float*>(wsb + pl.off_coef);
        float* cpart = reinterpret_cast<float*>(wsb + pl.off_cpart);
        void* mscratch = cpart;          // matrix-pipe passes: images + records first, the VALU post pass's records (if any) behind
        if (pl.cmat_pre && !pl.cmat_post) cpart = reinterpret_cast<float*>(wsb + pl.off_cpart + ((cond_mfma_scratch_bytes(d->num_person, I, A) + 255) & ~(size_t)255));
        CondParams cp;
        memset(&cp, 0, sizeof(cp));
        const int cond_blocks = pl.cond_nblk;
        // fp32 rows + more than one pass: the first pass (cond_pre / row_count) leaves 1-byte cell codes of the minibatch's rows
        // in the workspace (already gathered), every later pass reads those
        const bool emit = pl.off_codes != 0 && !pl.given && response != nullptr;
        uint8_t* code_rows = emit ? reinterpret_cast<uint8_t*>(wsb + pl.off_codes) : nullptr;
        cp.codes_stride = pl.codes_stride;
        cp.response = response; cp.mask = mask; cp.row_index = row_index; cp.table = table;
        cp.resp_stride = d->response_row_stride; cp.mask_stride = d->mask_row_stride;
        cp.B = d->num_person; cp.I_total = I; cp.A = A; cp.mask_dtype = d->mask_dtype;
        cp.coef_panels = pl.panels; cp.rec_stride = pl.cond_rec; cp.coef_in = coef;
        e = hipSuccess;
        const bool given_direct = pl.given && pl.panels == 1;      // one panel: the kernel's slot lanes read / write the posterior themselves
        if (pl.cond_fused) {
            // no first pass: the matrix kernel gathers the experts itself (XM == 3) and writes the cell codes the gradient pass reads
            p.cond_table = table;
            p.codes_out = (emit && grad) ? code_rows : nullptr;
            p.codes_stride = pl.codes_stride;
        } else if (given_direct) {
            p.given_post = table;
            p.given_grad = grad ? grad_table : nullptr;
            p.table = item;               // the 2-row expert table is not used in this mode: any finite floats (>= 4 A of them)
        } else if (pl.given) {
            const long long n = (long long)d->num_person * (A + 1);
            hipLaunchKernelGGL(given_pre_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, pre,
                               (long long)d->num_person, A, I);
            e = hipGetLastError();
            p.pre_stats = pre;
            p.pre_panels = 1;
            p.table = item;               // the 2-row expert table is not used in this mode: any finite floats (>= 4 A of them)
        } else if (pl.cmat_pre) {
            // matrix-pipe pre pass.  fp32 rows: the contraction kernel reads them itself and leaves the rows' cell codes behind for
            // the passes that follow (minibatch order) -- round 3 ran a re-pack stream (row_count_kernel) in front of it
            if (emit) {
                e = launch_cond_pre_mfma_fp32(response, d->mask_dtype == VIBO_MASK_U8 ? mask : nullptr, (long long)d->response_row_stride,
                                              (long long)d->mask_row_stride, row_index, d->num_person, I, A, table, pre, code_rows,
                                              (long long)pl.codes_stride, mscratch, s);
            } else {
                e = launch_cond_pre_mfma(static_cast<const uint8_t*>(mask), d->mask_row_stride, row_index, d->num_person, I, A, table, pre,
                                         mscratch, s);
            }
            p.pre_stats = pre;
            p.pre_panels = 1;
        } else if (pl.cond) {
            // more than one panel: ALL of them in one launch per 4 ability dims (CondParams::panel_count), the chip's workgroup
            // slots shared out over the panels
            const bool one_launch = pl.panels > 1;
            const int pre_launches = one_launch ? 1 : pl.panels;
            int pre_blocks = cond_blocks;
            if (one_launch) {
                int per = cond_blocks / pl.panels;
                if (per < 1) per = 1;
                pre_blocks = per * pl.panels;
                cp.panel_count = pl.panels;
            }
            for (int pn = 0; pn < pre_launches && e == hipSuccess; ++pn) {
                cp.item0 = pn * 1024;
                cp.I = one_launch ? 1024 : (I - cp.item0 < 1024 ? I - cp.item0 : 1024);
                cp.pre_out = pre + (size_t)pn * d->num_person * (2 * A + 1);
                for (cp.a0 = 0; cp.a0 < A && e == hipSuccess; cp.a0 += 4) {     // 4 ability dims per launch
                    cp.codes_out = (emit && cp.a0 == 0) ? code_rows : nullptr;
                    CondParams cq = cp;
                    if (emit && cp.a0 > 0) {      // dims 4..7: the rows' cell codes are there already (written by the first launch)
                        cq.response = nullptr; cq.mask = code_rows; cq.row_index = nullptr;
                        cq.mask_stride = pl.codes_stride; cq.mask_dtype = VIBO_MASK_CODES;
                    }
                    e = launch_cond_pre(cq, A == 1 ? 1 : A <= 2 ? 2 : 4, (cp.I + 255) / 256, pre_blocks, s);   // own template width (3PL widens the split kernel's)
                }
            }
            cp.codes_out = nullptr;
            cp.panel_count = 0;
            p.pre_stats = pre;
            p.pre_panels = pl.panels;
            if (pl.panels > 1 && e == hipSuccess) {
                const long long n = (long long)d->num_person * (2 * A + 1);
                hipLaunchKernelGGL(panel_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, n, pl.panels);
                e = hipGetLastError();
                p.pre_panels = 1;
            }
        } else {
            int* cnt = reinterpret_cast<int*>(wsb + pl.off_cnt);
            if (row_counts && !row_index) {
                // the caller's whole-row counts (vibo_row_counts of the same rows, kept with its resident data: they depend on the
                // data alone): no count pass -- half of the call on rows of more than 1024 items (100k x 10k: 1.64 -> 0.85 ms)
                p.row_cnt = row_counts;
            } else if (row_counts) {
                hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)((d->num_person + 255) / 256)), dim3(256), 0, s, row_counts, row_index,
                                   cnt, d->num_person);
                e = hipGetLastError();
                p.row_cnt = cnt;
            } else {
                int cgrid = num_cu * 8;
                if (cgrid > (d->num_person + 3) / 4) cgrid = (d->num_person + 3) / 4;
                hipLaunchKernelGGL(row_count_kernel, dim3(cgrid), dim3(256), 0, s, response, mask, row_index, cnt,
                                   (long long)d->response_row_stride, (long long)d->mask_row_stride, d->num_person, I, d->mask_dtype,
                                   code_rows, (long long)pl.codes_stride);
                e = hipGetLastError();
                p.row_cnt = cnt;
            }
        }
        if (emit && pl.cond_fused) {  // (the matrix kernel reads the fp32 rows; the gradient pass the codes it leaves behind)
            cp.response = nullptr; cp.mask = code_rows; cp.row_index = nullptr;
            cp.mask_stride = pl.codes_stride; cp.mask_dtype = VIBO_MASK_CODES;
        } else if (emit) {            // from here on the rows are the cell codes just written (in minibatch order)
            codes = true;
            p.response = nullptr; p.mask = code_rows; p.row_index = nullptr;
            p.mask_stride = pl.codes_stride; p.mask_dtype = VIBO_MASK_CODES;
            cp.response = nullptr; cp.mask = code_rows; cp.row_index = nullptr;
            cp.mask_stride = pl.codes_stride; cp.mask_dtype = VIBO_MASK_CODES;
        }
        if (pl.msplit && pl.panels > 1 && e == hipSuccess) {
            // the matrix kernel takes all panels in one launch: workgroup = (panel, slot), see ElboParams::panel_count
            p.item0 = 0; p.I = 1024; p.primary = 1; p.panel_count = pl.panels;
            p.partial = partial;
            p.post_coef = ((pl.cond || (pl.given && !given_direct)) && grad) ? coef : nullptr;
            e = launch_split(p, pl.AT, codes, d->irt_model, grad, 4, pl.panels * pl.split_nblk, s, true);
        } else
        for (int pn = 0; pn < pl.panels && e == hipSuccess; ++pn) {
            p.item0 = pn * 1024;
            p.I = I - p.item0 < 1024 ? I - p.item0 : 1024;
            p.primary = pn == 0 ? 1 : 0;
            p.partial = partial + (size_t)pn * pl.split_nblk * pl.lay.stride;
            p.post_coef = ((pl.cond || (pl.given && !given_direct)) && grad) ? coef + (size_t)pn * d->num_person * 4 * A : nullptr;
            const int nq = (p.I + 255) / 256;
            if (pl.cond_fused)
                e = p.row_index ? launch_elbo_msplit_xg(p, d->irt_model, grad, (p.I + 127) / 128, pl.split_nblk, s)
                                : launch_elbo_msplit_xa(p, d->irt_model, grad, (p.I + 127) / 128, pl.split_nblk, s);
            else
                e = launch_split(p, pl.AT, codes, d->irt_model, grad, nq, pl.split_nblk, s, pl.msplit);
        }
        if (pl.cond && grad) {
            if (pl.panels > 1 && e == hipSuccess) {       // the panels' backward coefficients summed once (cond_post reads 1 block, not `panels`)
                const long long n = (long long)d->num_person * 4 * A;
                hipLaunchKernelGGL(panel_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, coef, n, pl.panels);
                e = hipGetLastError();
                cp.coef_panels = 1;
            }
            if (pl.cmat_post) {
                if (e == hipSuccess)
                    e = launch_cond_post_mfma(static_cast<const uint8_t*>(cp.mask), cp.mask_stride, cp.row_index, d->num_person, I, A, table,
                                              coef, grad_table, mscratch, s, &tail);
            } else {
                for (int pn = 0; pn < pl.panels && e == hipSuccess; ++pn) {
                    cp.item0 = pn * 1024;
                    cp.I = I - cp.item0 < 1024 ? I - cp.item0 : 1024;
                    cp.partial = cpart + (size_t)pn * pl.cond_post_nblk * pl.cond_rec;
                    for (cp.a0 = 0; cp.a0 < A && e == hipSuccess; cp.a0 += 4)
                        e = launch_cond_post(cp, A == 1 ? 1 : A <= 2 ? 2 : 4, (cp.I + 255) / 256, pl.cond_post_nblk, s);
                }
                if (e == hipSuccess) e = launch_cond_finalize(cpart, grad_table, I, A, pl.panels, pl.cond_post_nblk, pl.cond_rec, s, &tail);
            }
        }
        if (pl.given && !given_direct && grad && e == hipSuccess) {
            const long long n = (long long)d->num_person * A;
            hipLaunchKernelGGL(given_post_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, coef, pl.panels,
                               grad_table, (long long)d->num_person, A);
            e = hipGetLastError();
        }
        nblk_used = pl.panels * pl.split_nblk;
        panel_items = 1024;
        bpp = pl.split_nblk;
    } else if (pl.split_ok && vec) {
        nblk_used = pl.split_nblk;
        if (pl.narrow) e = launch_elbo_narrow(p, codes, d->irt_model, grad, pl.split_nblk, s);
        else e = launch_split(p, pl.AT, codes, d->irt_model, grad, pl.split_nq, pl.split_nblk, s, pl.msplit);
    } else if (pl.row_ok && vec && I % 4 == 0 && pl.AT == A) {
        nblk_used = pl.row_nblk;
        e = launch_elbo_rows(p, d->irt_model, grad, pl.row_nblk, s);
    } else
    switch (pl.AT) {
        case 1: e = launch_elbo_a1(p, d->irt_model, grad, pl.geom, s); break;
        case 2: e = launch_elbo_a2(p, d->irt_model, grad, pl.geom, s); break;
        case 4: e = launch_elbo_a4(p, d->irt_model, grad, pl.geom, s); break;
        default: e = launch_elbo_a8(p, d->irt_model, grad, pl.geom, s); break;
    }
    if (e != hipSuccess) return hip_fail(e, "elbo kernel launch");
    if (skip_finalize) return 0;      // the partial records stay in the workspace for vibo_train_epilogue_fused

    FinalizeParams f;
    memset(&f, 0, sizeof(f));
    f.partial = partial; f.out_scalars = out_scalars; f.grad_table = (pl.panels > 0 && (pl.cond || pl.given)) ? nullptr : grad_table; f.grad_item = grad_item;
    f.grad_flow = grad_flow;
    f.nblk = nblk_used; f.I = I; f.A = A; f.D = pl.D; f.n_flows = d->n_flows; f.reg_mode = d->reg_mode;
    f.irt = d->irt_model; f.want_grad = grad ? 1 : 0; f.lay = pl.lay;
    f.panel_items = panel_items; f.bpp = bpp ? bpp : nblk_used;
    const int n_out = 8 + (grad ? 8 * A + 2 * d->n_flows * (2 * A + 1) + I * pl.D : 0);
    // (the conditional posterior's table-gradient finalize rides in the same launch: workgroups past n_fin)
    f.tail = tail;
    const int n_tail = tail.kind ? tail.gx * tail.gy : 0;
    if (f.bpp >= 1024 || n_out <= 64) {    // many small records, or the 8 scalars of a forward-only call: more slices per output
        f.n_fin = (n_out + 15) / 16;
        hipLaunchKernelGGL(finalize_kernel<16>, dim3(f.n_fin + n_tail), dim3(1024), 0, s, f);
    } else {
        f.n_fin = (n_out + 63) / 64;
        hipLaunchKernelGGL(finalize_kernel<64>, dim3(f.n_fin + n_tail), dim3(1024), 0, s, f);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "finalize launch");
    return 0;
}

extern "C" {

int vibo_elbo_fwd_bwd(const vibo_desc* d, const float* response, const void* mask, const int64_t* row_index,
                      const float* table, const float* item, const float* eps, const float* flow,
                      float* out_scalars, float* ability_mu, float* ability_logvar, float* ability,
                      float* ability_k, float* ability_ladj, float* grad_table, float* grad_item,
                      float* grad_flow, void* workspace, size_t workspace_bytes, void* stream) {
    return elbo_fwd_bwd_impl(d, nullptr, 0, response, mask, row_index, table, item, eps, flow, out_scalars, ability_mu, ability_logvar,
                             ability, ability_k, ability_ladj, grad_table, grad_item, grad_flow, workspace, workspace_bytes, stream);
}

int vibo_elbo_fwd_bwd_counts(const vibo_desc* d, const float* response, const void* mask, const int64_t* row_index, const int32_t* row_counts,
                             const float* table, const float* item, const float* eps, const float* flow,
                             float* out_scalars, float* ability_mu, float* ability_logvar, float* ability,
                             float* ability_k, float* ability_ladj, float* grad_table, float* grad_item,
                             float* grad_flow, void* workspace, size_t workspace_bytes, void* stream) {
    if (!row_counts) return fail(-5, "null row_counts");
    return elbo_fwd_bwd_impl(d, nullptr, 0, response, mask, row_index, table, item, eps, flow, out_scalars, ability_mu, ability_logvar,
                             ability, ability_k, ability_ladj, grad_table, grad_item, grad_flow, workspace, workspace_bytes, stream, row_counts);
}

int vibo_train_step_supported(const vibo_desc* d) {
    if (check_desc(d) != 0) return 0;
    Plan pl;
    if (make_plan(d, &pl) < 0) return 0;
    if (!step_plan_ok(d, pl)) return 0;
    // bit 1: vibo_train_epilogue_fused can also take over the finalize -- where it sums the partial records in the same order as the
    // stand-alone finalize would (16 slices; elbo_fwd_bwd_impl picks the 64-slice finalize_kernel<16> for fewer than 65 outputs
    // or 1024+ records: there the four-launch form and the folded one would differ in the last bits)
    const int n_out = 8 + 8 * d->ability_dim + d->num_item * pl.D;
    return 1 | ((pl.split_nblk < 1024 && n_out > 64) ? 2 : 0);
}

int vibo_elbo_fwd_bwd_step(const vibo_desc* d, int32_t* step_count, int skip_finalize, const float* response, const void* mask,
                           const int64_t* row_index, const float* table, const float* item, const float* eps, float* out_scalars,
                           float* ability_mu, float* ability_logvar, float* ability, float* grad_table, float* grad_item,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (!step_count) return fail(-5, "null step_count");
    return elbo_fwd_bwd_impl(d, step_count, skip_finalize, response, mask, row_index, table, item, eps, nullptr, out_scalars, ability_mu,
                             ability_logvar, ability, nullptr, nullptr, grad_table, grad_item, nullptr, workspace, workspace_bytes, stream);
}

int vibo_train_epilogue_fused(const vibo_desc* d, int hidden_dim, const void* workspace, float* flat, float* saved_h,
                              float* kl_parts, float* eps_item, const float* beta, const float* lr, int32_t* step_count,
                              float* mlp_params, float* mlp_m, float* mlp_v, float* item_mu, float* item_logvar, float* item_m,
                              float* item_v, float* loss_out, uint64_t seed, float* item_feat, float* table, float* eps_ability,
                              int64_t n_eps_ability, uint32_t ability_stream_id, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (hidden_dim < 1 || hidden_dim > kMaxHidden) return fail(-6, "hidden_dim outside 1..%d", kMaxHidden);
    if (d->posterior != VIBO_POSTERIOR_UNCONDITIONAL || d->n_flows != 0 || d->reg_mode != VIBO_REG_KL)
        return fail(-6, "vibo_train_epilogue_fused: plain model only (unconditional posterior, no flows, KL regulariser)");
    if (!flat || !saved_h || !kl_parts || !eps_item || !beta || !lr || !step_count || !mlp_params || !mlp_m || !mlp_v || !item_mu ||
        !item_logvar || !item_m || !item_v || !loss_out || !item_feat || !table || !eps_ability || n_eps_ability < 0)
        return fail(-5, "null required pointer");
    EpiParams e;
    memset(&e, 0, sizeof(e));
    const int I = d->num_item, A = d->ability_dim, D = item_feat_dim(d->irt_model, A);
    e.H = hidden_dim; e.O = 2 * A; e.n_item_entries = I * D; e.I = I; e.D = D;
    e.flat_in = flat; e.flat_out = flat; e.saved_h = saved_h; e.kl_parts = kl_parts; e.eps_item = eps_item; e.beta = beta; e.lr = lr;
    e.step_count = step_count; e.P = mlp_params; e.M = mlp_m; e.V = mlp_v; e.mu = item_mu; e.lv = item_logvar; e.im = item_m;
    e.iv = item_v; e.loss_out = loss_out; e.item_feat = item_feat; e.table = table;
    if (workspace) {
        // the partial records vibo_elbo_fwd_bwd_step(skip_finalize) left behind: same descriptor -> same plan
        Plan pl;
        if (make_plan(d, &pl) < 0) return fail(-8, "no plan for this descriptor");
        if (!step_plan_ok(d, pl)) return fail(-8, "vibo_train_epilogue_fused: the descriptor is not a vibo_elbo_fwd_bwd_step call");
        if ((uintptr_t)workspace & 255) return fail(-7, "workspace must be 256-byte aligned");
        e.partial = reinterpret_cast<const float*>(static_cast<const char*>(workspace) + pl.off_partial);
        e.nblk = pl.split_nblk;
        e.lay = pl.lay;
        if (e.nblk >= 1024) return fail(-8, "vibo_train_epilogue_fused: %d partial records (finalize order of many small records): "
                                            "call vibo_elbo_fwd_bwd_step without skip_finalize", e.nblk);
    }
    e.seed_lo = (uint32_t)seed; e.seed_hi = (uint32_t)(seed >> 32);
    e.eps_ab = eps_ability; e.n_ab = (long long)n_eps_ability; e.ab_stream = ability_stream_id;
    e.n_item_blocks = train_epilogue_item_blocks(e.n_item_entries);
    const hipError_t he = launch_train_epilogue_fused(e, (hipStream_t)stream);
    if (he != hipSuccess) return hip_fail(he, "train_epilogue_fused launch");
    return 0;
}

// plan of a multi-sample forward: the single-launch plan with want_grad = 0, restricted to the row-split paths
static int multi_plan(const vibo_desc* d, vibo_desc* d0, Plan* pl, size_t* prep_bytes) {
    *d0 = *d;
    d0->want_grad = 0;
    const int stride = make_plan(d0, pl, false);
    if (stride < 0) return stride;
    // conditional posterior: the expert table itself depends on the item sample, nothing is shared between samples
    if (d->posterior == VIBO_POSTERIOR_CONDITIONAL) return fail(-8, "multi-sample forward: conditional posterior (one table per sample)");
    if (d->posterior == VIBO_POSTERIOR_GIVEN) return fail(-8, "multi-sample forward: caller-supplied posterior");
    if (pl->general || !(pl->split_ok || pl->panels > 0)) return fail(-8, "multi-sample forward: shape is not on the row-split path");
    *prep_bytes = ((size_t)((d->num_item + 15) & ~15) * pl->DP * 4 + 255) & ~(size_t)255;
    return 0;
}

size_t vibo_multi_workspace_bytes(const vibo_desc* d, int num_samples) {
    if (check_desc(d) != 0 || num_samples < 1) return 0;
    vibo_desc d0;
    Plan pl;
    size_t prep = 0;
    if (multi_plan(d, &d0, &pl, &prep) != 0) return 0;
    return pl.total_bytes + 4 * prep;
}

int vibo_elbo_multi_forward(const vibo_desc* d, int num_samples, const float* response, const void* mask,
                            const int64_t* row_index, const float* table, const float* item, const float* eps,
                            const float* flow, float* out_scalars, void* workspace, size_t workspace_bytes, void* stream) {
    const int num_cu = device_cus();
    int rc = check_desc(d);
    if (rc) return rc;
    if (num_samples < 1) return fail(-3, "num_samples must be >= 1");
    if ((!response && d->mask_dtype != VIBO_MASK_CODES) || !table || !item || !eps || !out_scalars) return fail(-5, "null required pointer");
    if ((d->mask_dtype == VIBO_MASK_NONE) != (mask == nullptr)) return fail(-5, "mask pointer / mask_dtype mismatch");
    if (d->n_flows > 0 && !flow) return fail(-5, "flows need flow");
    vibo_desc d0;
    Plan pl;
    size_t prep = 0;
    rc = multi_plan(d, &d0, &pl, &prep);
    if (rc) return rc;
    if (!workspace || workspace_bytes < pl.total_bytes + 4 * prep) return fail(-7, "workspace too small");
    if ((uintptr_t)workspace & 255) return fail(-7, "workspace must be 256-byte aligned");
    const int I = d->num_item, A = d->ability_dim;
    if (!rows_vec_ok(d, response, mask)) return fail(-8, "multi-sample forward: rows are not 16-byte chunkable");
    hipStream_t s = (hipStream_t)stream;
    char* wsb = static_cast<char*>(workspace);
    float* partial = reinterpret_cast<float*>(wsb + pl.off_partial);
    float* item_prep = reinterpret_cast<float*>(wsb + pl.total_bytes);       // up to 4 prepped tables
    const int panels = pl.panels > 0 ? pl.panels : 1;
    const int nblk = pl.split_nblk;

    MultiParams mp;
    memset(&mp, 0, sizeof(mp));
    ElboParams& p = mp.e;
    p.response = response; p.mask = mask; p.row_index = row_index; p.table = table; p.item_prep = item_prep;
    p.resp_stride = d->response_row_stride; p.mask_stride = d->mask_row_stride;
    p.B = d->num_person; p.I = I; p.A = A; p.D = pl.D; p.DP = pl.DP;
    p.mask_dtype = d->mask_dtype; p.missing_mode = d->missing_mode; p.reg_mode = d->reg_mode;
    p.flow = flow; p.n_flows = d->n_flows; p.lay = pl.lay; p.I_total = I; p.primary = 1;
    mp.item_sstride = (long long)(prep / 4);
    mp.eps_sstride = (long long)d->num_person * A;
    hipError_t e = hipSuccess;
    if (pl.panels > 0) {                       // sample-independent: whole-row counts
        int* cnt = reinterpret_cast<int*>(wsb + pl.off_cnt);
        int cgrid = num_cu * 8;
        if (cgrid > (d->num_person + 3) / 4) cgrid = (d->num_person + 3) / 4;
        hipLaunchKernelGGL(row_count_kernel, dim3(cgrid), dim3(256), 0, s, response, mask, row_index, cnt,
                           (long long)d->response_row_stride, (long long)d->mask_row_stride, d->num_person, I, d->mask_dtype);
        e = hipGetLastError();
        p.row_cnt = cnt;
    }
    if (e != hipSuccess) return hip_fail(e, "multi-sample forward: pre-pass launch");
    const int sc_max = pl.AT <= 4 ? 4 : 2;
    for (int s0 = 0; s0 < num_samples;) {
        const int rem = num_samples - s0;
        const int sc = rem >= 4 && sc_max >= 4 ? 4 : rem >= 2 ? 2 : 1;
        for (int k = 0; k < sc; ++k)
            hipLaunchKernelGGL(item_prep_kernel, dim3((I + 15 + 255) / 256), dim3(256), 0, s, item + (size_t)(s0 + k) * I * pl.D,
                               item_prep + (size_t)k * (prep / 4), I, A, pl.AT, pl.D, pl.DP, d->irt_model);
        p.eps = eps + (size_t)s0 * d->num_person * A;
        for (int pn = 0; pn < panels && e == hipSuccess; ++pn) {
            p.item0 = pn * 1024;
            p.I = pl.panels > 0 ? (I - p.item0 < 1024 ? I - p.item0 : 1024) : I;
            p.primary = pn == 0 ? 1 : 0;
            p.partial = partial + (size_t)pn * nblk * pl.lay.stride;
            e = launch_elbo_multi(mp, pl.AT, d->irt_model, sc, (p.I + 255) / 256, nblk, s);
        }
        if (e != hipSuccess) return hip_fail(e, "multi-sample forward launch");
        hipLaunchKernelGGL(multi_finalize_kernel, dim3(1), dim3(1024), 0, s, partial, out_scalars + (size_t)s0 * VIBO_NUM_SCALARS,
                           panels * nblk, pl.lay.stride, sc, d->reg_mode);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "multi-sample finalize launch");
        s0 += sc;
    }
    return 0;
}

int vibo_decode_mean(const vibo_desc* d, int num_samples, const float* ability, const float* item,
                     float* response_mu_mean, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (num_samples < 1) return fail(-3, "num_samples must be >= 1");
    if (!ability || !item || !response_mu_mean) return fail(-5, "null required pointer");
    const int I = d->num_item, A = d->ability_dim, D = item_feat_dim(d->irt_model, A);
    const long long B = d->num_person;
    const long long by = (B + 7) / 8;
    if (by > 65535LL * 32768) return fail(-3, "num_person too large");
    // grid.y is limited to 65535: loop in chunks of 65535 * 8 persons
    for (long long y0 = 0; y0 < by; y0 += 65535) {
        const int ny = (int)((by - y0 < 65535) ? (by - y0) : 65535);
        const long long p0 = y0 * 8;
        const int nb = (int)((B - p0 < (long long)ny * 8) ? (B - p0) : (long long)ny * 8);
        // ability rows of sample s start at ability + s * B * A: pass the full B as the sample stride via a shifted base
        hipLaunchKernelGGL(decode_mean_kernel_strided, dim3((I + 255) / 256, ny), dim3(256), 0, (hipStream_t)stream,
                           ability + p0 * A, item, response_mu_mean + p0 * I, num_samples, nb, (int)B, I, A, D, d->irt_model);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "decode_mean launch");
    return 0;
}

int vibo_encode(const vibo_desc* d, const float* response, const void* mask, const int64_t* row_index,
                const float* table, float* ability_mu, float* ability_logvar, void* workspace,
                size_t workspace_bytes, void* stream) {
    const int num_cu = device_cus();
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->posterior == VIBO_POSTERIOR_GIVEN) return fail(-3, "vibo_encode: the posterior is the caller's own with VIBO_POSTERIOR_GIVEN");
    if ((!response && d->mask_dtype != VIBO_MASK_CODES) || !table || !ability_mu || !ability_logvar) return fail(-5, "null required pointer");
    if ((d->mask_dtype == VIBO_MASK_NONE) != (mask == nullptr)) return fail(-5, "mask pointer / mask_dtype mismatch");
    {
        // fast path: the row statistics of the row-split pipeline (16-byte row chunks at HBM speed) + a per-person finish
        const int I = d->num_item, A = d->ability_dim;
        const size_t need = encode_scratch_bytes(d);
        const bool vec = need > 0 && rows_vec_ok(d, response, mask);
        if (vec && workspace && workspace_bytes >= need && (((uintptr_t)workspace & 255) == 0)) {
            hipStream_t s = (hipStream_t)stream;
            const long long BA = (long long)d->num_person * A;
            hipError_t e = hipSuccess;
            if (d->posterior == VIBO_POSTERIOR_CONDITIONAL) {
                float* pre = static_cast<float*>(workspace);
                const int panels = (I + 1023) / 1024;
                int grid = num_cu * 3;
                if (grid > (d->num_person + 7) / 8) grid = (d->num_person + 7) / 8;
                CondParams cp;
                memset(&cp, 0, sizeof(cp));
                cp.response = response; cp.mask = mask; cp.row_index = row_index; cp.table = table;
                cp.resp_stride = d->response_row_stride; cp.mask_stride = d->mask_row_stride;
                cp.B = d->num_person; cp.I_total = I; cp.A = A; cp.mask_dtype = d->mask_dtype;
                const bool mfma = encode_on_matrix_pipe(d);
                if (mfma) {
                    const size_t pre_bytes = ((size_t)panels * d->num_person * (2 * A + 1) * 4 + 255) & ~(size_t)255;
                    e = launch_cond_pre_mfma(static_cast<const uint8_t*>(mask), d->mask_row_stride, row_index, d->num_person, I, A, table, pre,
                                             static_cast<char*>(workspace) + pre_bytes, s);
                }
                for (int pn = 0; pn < panels && e == hipSuccess && !mfma; ++pn) {
                    cp.item0 = pn * 1024;
                    cp.I = I - cp.item0 < 1024 ? I - cp.item0 : 1024;
                    cp.pre_out = pre + (size_t)pn * d->num_person * (2 * A + 1);
                    for (cp.a0 = 0; cp.a0 < A && e == hipSuccess; cp.a0 += 4)
                        e = launch_cond_pre(cp, A == 1 ? 1 : A <= 2 ? 2 : 4, (cp.I + 255) / 256, grid, s);
                }
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(encode_finish_kernel, dim3((unsigned)((BA + 255) / 256)), dim3(256), 0, s, nullptr, pre, mfma ? 1 : panels,
                                       table, ability_mu, ability_logvar, (long long)d->num_person, I, A, d->missing_mode);
                    e = hipGetLastError();
                }
            } else {
                int* cnt = static_cast<int*>(workspace);
                int cgrid = num_cu * 8;
                if (cgrid > (d->num_person + 3) / 4) cgrid = (d->num_person + 3) / 4;
                hipLaunchKernelGGL(row_count_kernel, dim3(cgrid), dim3(256), 0, s, response, mask, row_index, cnt,
                                   (long long)d->response_row_stride, (long long)d->mask_row_stride, d->num_person, I, d->mask_dtype);
                hipLaunchKernelGGL(encode_finish_kernel, dim3((unsigned)((BA + 255) / 256)), dim3(256), 0, s, cnt, nullptr, 0, table,
                                   ability_mu, ability_logvar, (long long)d->num_person, I, A, d->missing_mode);
                e = hipGetLastError();
            }
            if (e != hipSuccess) return hip_fail(e, "encode (fast path) launch");
            return 0;
        }
    }
    if (d->mask_dtype == VIBO_MASK_CODES) return codes_unsupported();      // (or the workspace is missing / too small)
    EncodeParams p;
    memset(&p, 0, sizeof(p));
    p.response = response; p.mask = mask; p.row_index = row_index; p.table = table;
    p.ability_mu = ability_mu; p.ability_logvar = ability_logvar;
    p.resp_stride = d->response_row_stride; p.mask_stride = d->mask_row_stride;
    p.B = d->num_person; p.I = d->num_item; p.A = d->ability_dim;
    p.mask_dtype = d->mask_dtype; p.missing_mode = d->missing_mode;
    p.conditional = d->posterior == VIBO_POSTERIOR_CONDITIONAL;
    hipLaunchKernelGGL(encode_kernel, dim3((d->num_person + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "encode launch");
    return 0;
}

int vibo_row_counts(const vibo_desc* d, const float* response, const void* mask, const int64_t* row_index, int32_t* counts,
                    void* stream) {
    const int num_cu = device_cus();
    int rc = check_desc(d);
    if (rc) return rc;
    if ((!response && d->mask_dtype != VIBO_MASK_CODES) || !counts) return fail(-5, "null required pointer");
    if ((d->mask_dtype == VIBO_MASK_NONE) != (mask == nullptr)) return fail(-5, "mask pointer / mask_dtype mismatch");
    if (d->num_item > 32767) return fail(-3, "vibo_row_counts: the packed counts hold up to 32767 items");
    hipStream_t s = (hipStream_t)stream;
    if (d->num_item >= 4 && d->mask_dtype != VIBO_MASK_I64 && rows_vec_ok(d, response, mask)) {
        int cgrid = num_cu * 8;
        if (cgrid > (d->num_person + 3) / 4) cgrid = (d->num_person + 3) / 4;
        hipLaunchKernelGGL(row_count_kernel, dim3(cgrid), dim3(256), 0, s, response, mask, row_index, counts,
                           (long long)d->response_row_stride, (long long)d->mask_row_stride, d->num_person, d->num_item, d->mask_dtype);
    } else {
        hipLaunchKernelGGL(row_count_scalar_kernel, dim3((d->num_person + 3) / 4), dim3(256), 0, s, response, mask, row_index, counts,
                           (long long)d->response_row_stride, (long long)d->mask_row_stride, d->num_person, d->num_item, d->mask_dtype);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "row_counts launch");
    return 0;
}

int vibo_pack_codes(const vibo_desc* d, const float* response, const void* mask, uint8_t* codes, int64_t codes_row_stride,
                    void* stream) {
    // fast path: aligned rows, 4 cells per thread (16 B of responses + 4 B of mask -> one code word)
    if (check_desc(d) == 0 && response && codes && d->mask_dtype != VIBO_MASK_CODES && d->mask_dtype != VIBO_MASK_I64 &&
        (d->mask_dtype == VIBO_MASK_NONE) == (mask == nullptr) && codes_row_stride % 4 == 0 &&
        codes_row_stride >= ((d->num_item + 3) & ~3) && (((uintptr_t)codes & 3) == 0) && rows_vec_ok(d, response, mask)) {
        const long long n = (long long)d->num_person * (codes_row_stride / 4);
        long long grid = (n + 255) / 256;
        if (grid > 262144) grid = 262144;
        hipLaunchKernelGGL(pack_codes4_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, response, mask,
                           reinterpret_cast<uint32_t*>(codes), (long long)d->response_row_stride, (long long)d->mask_row_stride,
                           (long long)(codes_row_stride / 4), (long long)d->num_person, d->num_item, d->mask_dtype);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "pack_codes launch");
        return 0;
    }
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->mask_dtype == VIBO_MASK_CODES) return fail(-3, "vibo_pack_codes: the source rows are already cell codes");
    if (!response || !codes) return fail(-5, "null required pointer");
    if ((d->mask_dtype == VIBO_MASK_NONE) != (mask == nullptr)) return fail(-5, "mask pointer / mask_dtype mismatch");
    if (codes_row_stride < d->num_item) return fail(-3, "codes_row_stride %lld < num_item", (long long)codes_row_stride);
    const long long n = (long long)d->num_person * codes_row_stride;
    long long grid = (n + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, response, mask, codes,
                       (long long)d->response_row_stride, (long long)d->mask_row_stride, (long long)codes_row_stride,
                       (long long)d->num_person, d->num_item, d->mask_dtype);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pack_codes launch");
    return 0;
}

int vibo_decode(const vibo_desc* d, const float* ability, const float* item, float* response_mu, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!ability || !item || !response_mu) return fail(-5, "null required pointer");
    if (d->num_person > 65535 * 1024) return fail(-3, "num_person too large for decode grid");
    const int I = d->num_item, A = d->ability_dim, D = item_feat_dim(d->irt_model, A);
    // grid.y is limited to 65535: loop in chunks
    const long long B = d->num_person;
    for (long long b0 = 0; b0 < B; b0 += 65535) {
        const int nb = (int)((B - b0 < 65535) ? (B - b0) : 65535);
        hipLaunchKernelGGL(decode_kernel, dim3((I + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream,
                           ability + b0 * A, item, response_mu + b0 * I, nb, I, A, D, d->irt_model);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "decode launch");
    return 0;
}

}  // extern "C"
