/*
 * rr_hip.h -- C ABI of librr_hip.so, the MI355X (gfx950) Muskingum routing engine.
 *
 * Drop-in boundary: these entry points replace the reference's kernel boundary, i.e. the three
 * numba functions and the UH convolution that the reference's routers call once per input file
 * (citations relative to the reference repository root):
 *
 *   rr_rapid_route      <- rapid_route      river_route/routers/_numba_kernels.py:49-84
 *                          (call site river_route/routers/RapidMuskingum.py:27-32)
 *   rr_muskingum_route  <- muskingum_route  river_route/routers/_numba_kernels.py:8-46
 *                          (call site river_route/routers/Muskingum.py:281-286)
 *   rr_unit_route       <- unit_route       river_route/routers/_numba_kernels.py:88-171
 *                          (call site river_route/routers/UnitMuskingum.py:82-92)
 *   rr_uh_convolve      <- UnitHydrograph.convolve  river_route/uhkernels/UnitHydrograph.py:77-107
 *                          (call site river_route/routers/UnitMuskingum.py:75)
 *   rr_plan_create      <- the CSC structure the routers take from tools.adjacency_matrix
 *                          (river_route/tools.py:75-109; river_route/routers/Muskingum.py:189-192)
 *   rr_plan_set_coeffs  <- the coefficient vectors of Muskingum._set_muskingum_coefficients
 *                          (river_route/routers/Muskingum.py:172-193) and c4_dt of RapidMuskingum.py:24
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is IEEE fp64; CSC indices are int32
 *     (what scipy hands the reference); hw/inner positions are implied by the adjacency.
 *   - every function returns RR_OK (0) or a negative RR_E_* code and never throws; the message of
 *     the last failure on the calling thread is rr_last_error().
 *   - arrays are C-contiguous; 2-D arrays are (time, reach) row-major in PARAMS-FILE reach order,
 *     exactly as the reference passes them.  The engine keeps its own permuted device layout.
 *   - `*_dev` variants take DEVICE pointers (valid on the plan's GPU) and a hipStream_t passed as
 *     void*; they only enqueue work: nothing in them allocates or synchronises (work memory comes
 *     from rr_plan_reserve -- a BREAK against the first versions of this ABI, see INTEGRATION.md: a
 *     `_dev` call or rr_stream_begin* without a reservation at least as large returns RR_E_STATE).  Every
 *     kernel of a call is enqueued on the caller's stream.  The un-suffixed variants take HOST pointers,
 *     copy in/out, and return when the results are in the caller's buffers.
 *   - a plan is bound to one GPU and is not thread-safe; use one plan per thread/stream.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute call fails with
 *     RR_E_NO_DEVICE.  rr_plan_create(device = RR_DEVICE_NONE) builds a host-only plan whose
 *     layout can be inspected (rr_plan_info / rr_plan_layout) but which cannot compute.
 */
#ifndef RR_HIP_H
#define RR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_OK 0
#define RR_E_INVALID (-1)         /* bad argument (null pointer, negative size, malformed CSC) */
#define RR_E_NOT_TOPOLOGICAL (-2) /* a CSC entry has row <= column: reaches not sorted upstream -> downstream */
#define RR_E_HIP (-3)             /* HIP runtime failure; text in rr_last_error() */
#define RR_E_NO_DEVICE (-4)       /* no usable GPU, or host-only plan asked to compute */
#define RR_E_STATE (-5)           /* call order (e.g. route before rr_plan_set_coeffs) */
#define RR_E_ALLOC (-6)           /* host or device allocation failed */
#define RR_E_UNSUPPORTED (-7)     /* structure outside what the engine handles */

#define RR_DEVICE_NONE (-1)

typedef struct rr_plan rr_plan;

/* ---- library ---- */
int rr_version(void);             /* major*10000 + minor*100 + patch */
const char *rr_last_error(void);  /* thread-local, never NULL */
int rr_device_count(void);        /* number of visible HIP devices, 0 when there is none */

/* ---- plan: network structure analysis + device-resident layout ---- */

/* n reaches; CSC of A[down, up] = 1 with n columns: csc_indptr[n+1], csc_indices[csc_indptr[n]].
 * Validates what tools.adjacency_matrix guarantees (row > column for every entry).  device is a
 * HIP device ordinal or RR_DEVICE_NONE. */
int rr_plan_create(int64_t n, const int32_t *csc_indptr, const int32_t *csc_indices, int device, rr_plan **out);
void rr_plan_destroy(rr_plan *plan);

/* info[0]=n, [1]=edges, [2]=depth (reaches on the longest flow path), [3]=widest level,
 * [4]=headwaters, [5]=outlets, [6]=1 if the engine's order equals params order (no permutation pass),
 * [7]=device ordinal or -1. */
int rr_plan_info(const rr_plan *plan, int64_t info[8]);

/* Engine layout for inspection/tests (any pointer may be NULL): perm[n] params index held at each
 * engine position; lag[n] pipeline lag of each position (ticks behind the farthest headwater);
 * child_ptr[n+1] engine-position range [child_ptr[p], child_ptr[p+1]) of the reaches flowing into p. */
int rr_plan_layout(const rr_plan *plan, int32_t *perm, int32_t *lag, int32_t *child_ptr);

/* Layout of the time-tiled kernel (subtree tiles), for inspection/tests.
 * info[0]=1 if the network tiles (else the streaming kernel routes it), [1]=tile capacity in positions, [2]=positions
 * (reaches + ghosts), [3]=ghosts, [4]=tiles, [5]=tile levels, [6]=threads per workgroup, [7]=rows the passes between
 * params-order rows and the kernel's records move per launch (a cyclic discharge array with at least that many rows is never
 * written twice by one launch).
 * rr_plan_tile_layout (any pointer may be NULL): tile_ptr[tiles+1] first position of each tile; tile_level[tiles];
 * perm[np] params index of the reach at (or mirrored by) each position; lag[np] pipeline lag, bit 28 set on a ghost, bit 27 on
 * a reach that a ghost of another tile mirrors; cfirst[np] first upstream position; ccnt[np] upstream positions (low 16 bits)
 * and how many of them are headwaters (high 16 bits, they come first); xpos[np] position of the mirroring ghost / mirrored
 * reach, -1 elsewhere. */
int rr_plan_tile_info(const rr_plan *plan, int64_t info[8]);
int rr_plan_tile_layout(const rr_plan *plan, int32_t *tile_ptr, int32_t *tile_level, int32_t *perm, int32_t *lag,
                        int32_t *cfirst, uint32_t *ccnt, int32_t *xpos);

/* Coefficients in params order.  lhs_off_data[e] is the off-diagonal of (I - diag(c1) A) for CSC entry e,
 * i.e. -c1[row(e)] (Muskingum.py:192); c2, c3 per reach; c4_dt per reach or NULL (channel-only / unit). */
int rr_plan_set_coeffs(rr_plan *plan, const double *lhs_off_data, const double *c2, const double *c3,
                       const double *c4_dt);

/* UnitMuskingum with general edge data: the reference's unit_route multiplies by a_inner_data[j] / a_hw_data[j] and
 * subtracts lhs_off_data[j] q_ch (river_route/routers/_numba_kernels.py:126-139, 159-162); its own callers pass ones and
 * -c1[row], which is what rr_plan_set_coeffs alone describes.  For other values: c1[n] per reach (params order, any value
 * on headwaters) and a_data per entry of the plan's CSC structure (a_inner_data / a_hw_data of that edge);
 * lhs_off_data of rr_plan_set_coeffs is then read for the edges between two reaches that have upstream reaches only.
 * Such a plan routes UnitMuskingum with the streaming kernel.  (NULL, NULL) goes back to unit weights. */
int rr_plan_set_unit_weights(rr_plan *plan, const double *c1, const double *a_data);

/* Work memory of the route calls to come, allocated up front: the *_dev entry points and rr_stream_begin* only enqueue
 * work and return RR_E_STATE (with the byte count in rr_last_error()) when a call needs more than has been reserved; the
 * host-pointer entry points, which synchronise anyway, reserve by themselves.  Reserve once per plan for the largest
 * call (T runoff rows x nsub sub-steps; rr_muskingum_route*: T = num_output_steps, nsub = num_routing_per_output), after
 * rr_plan_set_coeffs and rr_plan_set_boundary / rr_plan_set_options: one call per input file is the reference's pattern
 * (river_route/routers/TransformMuskingum.py:108-148), and the routers reserve in _hook_before_route.  Memory only
 * grows; a smaller call fits a larger reservation.  host_rows: bit 0 also prepares the staging of the host-pointer entry
 * points (pinned buffers and device rings of the PCIe pipeline); RR_ROWS_NOT_PLAIN: the call does not hand over lateral rows
 * in a device array (fused convolution, gridded runoff: rr_unit_route_uh*_dev, rr_rapid_route_runoff_dev), so the direct row path
 * -- which rr_rapid_route*_dev and rr_stream_begin take where the params order numbers small subtrees contiguously, see
 * rr_plan_direct_info -- does not apply and the record ring is needed; RR_ROWS_F32_OUT: the call writes float32 rows
 * (rr_*_route_f32*_dev: the direct task is then a multiple of 128 rows); RR_ROWS_UH: the call is rr_unit_route_uh*_dev (runoff
 * depths + unit-hydrograph kernel): on the direct row path -- which rr_unit_route*_dev take too, with up to four sub-steps per row and no
 * boundary reaches -- the convolution runs as a pass of its own into T rows of work memory, reserved here.
 * info (may be NULL): [0] 2 = direct row path, 1 = time-tiled kernel, 0 = streaming kernel; [1] routing ticks (rows) per launch K; [2] chunks of the
 * record ring (16 ticks each); [3] bytes of routing work memory now held on the device; [4] bytes of device staging and
 * [5] of pinned host staging of the host-pointer path; [6] depth of the routing pipeline in ticks (network depth + tile
 * levels x K: a call's first output row leaves this many ticks after its first input row entered); [7] bytes of the record
 * ring (or streaming work rows) this shape needs. */
#define RR_MODE_RAPID 0
#define RR_MODE_MUSKINGUM 1
#define RR_MODE_UNIT 2
#define RR_ROWS_NOT_PLAIN 2
#define RR_ROWS_F32_OUT 4
#define RR_ROWS_UH 8
int rr_plan_reserve(rr_plan *plan, int mode, int64_t T, int64_t nsub, int host_rows, int64_t info[8]);

/* The direct row path: where the params order numbers every small subtree contiguously -- any depth-first
 * post-order does (a valid order for tools.adjacency_matrix, river_route/tools.py:103-104) -- RapidMuskingum calls with one
 * sub-step per row on float64 device rows (rr_rapid_route_dev, rr_stream_begin) are routed straight from and to the caller's rows
 * by column-range tiles: no record ring and no permutation pass for 95 % of the columns.
 * info: [0] 1 = applies to this plan; [1] direct tiles; [2] holes (columns of skeleton reaches); [3] outlets that feed the
 * skeleton; [4] positions, [5] tiles, [6] tile levels of the skeleton; [7] rows of the LDS window.  why (may be NULL): the
 * reason it does not apply.  rr_plan_direct_layout: tile arrays [tiles], per-column arrays [n]; any pointer may be NULL. */
int rr_plan_direct_info(const rr_plan *plan, int64_t info[8], char *why, int64_t why_cap);
int rr_plan_direct_layout(const rr_plan *plan, int32_t *tile_c0, int32_t *tile_nc, int32_t *tile_lag_lo, int32_t *tile_span,
                          int32_t *delay, int32_t *up3, int32_t *xinfo);
/* The kernels around the routing kernel in the last call, sampled like it (every fourth launch between HIP events on the call's
 * stream, while rr_plan_set_options(sample_every >= 16) is in force): aux[3 k + 0] launches, [3 k + 1] launches sampled,
 * [3 k + 2] their milliseconds, for k = 0 the in-pass (params-order rows -> records), 1 the out-pass, 2 the skeleton's routing
 * launches of the direct row path, 3 its out-pass over the holes. */
int rr_plan_profile_aux(rr_plan *plan, double aux[12]);
/* Which routing kernel the last call on this plan ran. */
#define RR_KERNEL_TICK 0
#define RR_KERNEL_TILE 1
#define RR_KERNEL_DIRECT 2
int rr_plan_last_kernel(const rr_plan *plan);

/* Byte order of the FLOAT32 rows of the calls to come (rr_*_f32in_dev's lateral / depth rows, rr_*_f32*_dev's discharge rows): non-zero =
 * big-endian, as a NetCDF-3 file stores them, so that a file's bytes go to the device and come back from it as they are (rr_rows_upload /
 * rr_rows_download) and the conversion costs one instruction per value in the kernels that read or write the rows anyway.  Default: native. */
int rr_plan_set_row_format(rr_plan *plan, int in32_big_endian, int out32_big_endian);

/* Tuning / measurement.  rows_per_chunk: time rows moved per permutation launch of the streaming kernel (default 16).
 * sample_every >= 16: HIP-event brackets on the call's stream around sampled routing launches (every fourth launch of the
 * time-tiled kernel; every sample_every-th tick of the streaming kernel opens a bracket of 16 launches); 0 switches it off. */
int rr_plan_set_options(rr_plan *plan, int64_t rows_per_chunk, int64_t sample_every);

/* prof[0]=routing launches of the last route call, [1]=routing ticks inside brackets, [2]=sum of the bracket
 * durations (ms), [3]/[4]=smallest/largest per-launch mean of a bracket (ms), [5]=reaches updated by the
 * bracketed launches,
 * [6]=ms between the first and the last routing-step launch of the call (permutation passes included),
 * [7]=reach-steps of the call, [8]=number of brackets, [9]=routing ticks per launch (K of the time-tiled
 * kernel, 1 for the streaming kernel).  Synchronises the plan's last stream. */
int rr_plan_profile(rr_plan *plan, double prof[10]);

/* ---- routing, host pointers (the reference's kernel boundary) ---- */

/* q_t[n] in: initial state, out: state after the last sub-step.  qlateral[T*n], discharge[T*n].
 * discharge[t, i] = max(mean over the nsub sub-steps of step t of q[i], 0). */
int rr_rapid_route(rr_plan *plan, double *q_t, const double *qlateral, double *discharge,
                   int64_t num_runoff_steps, int64_t num_substeps);

/* discharge[num_output_steps * n]; each output row averages num_routing_per_output sub-steps. */
int rr_muskingum_route(rr_plan *plan, double *q_t, double *discharge,
                       int64_t num_output_steps, int64_t num_routing_per_output);

/* Plan built on the FULL adjacency; headwaters are the reaches with no upstream entry, inner reaches the
 * rest, both in ascending params order (UnitMuskingum.py:41-44).  q_ch[n_inner], q_full[n_inner] in/out;
 * convolved_lateral[T*n], discharge[T*n] over all n reaches. */
int rr_unit_route(rr_plan *plan, double *q_ch, double *q_full, const double *convolved_lateral,
                  double *discharge, int64_t num_runoff_steps, int64_t num_substeps);

/* kernel[n_ks*n], state[n_ks*n] in/out (carry-over), lateral[T*n] -> out[T*n].  No plan needed. */
int rr_uh_convolve(int device, const double *kernel, double *state, const double *lateral, double *out,
                   int64_t T, int64_t n_ks, int64_t n);

/* ---- routing, device pointers + stream ---- */

/* As above with device pointers.  qlateral has ql_rows rows and step t reads row t % ql_rows;
 * discharge has out_rows rows and step t writes row t % out_rows (pass T for plain arrays). */
int rr_rapid_route_dev(rr_plan *plan, double *q_t, const double *qlateral, int64_t ql_rows,
                       double *discharge, int64_t out_rows, int64_t num_runoff_steps, int64_t num_substeps,
                       void *stream);
int rr_muskingum_route_dev(rr_plan *plan, double *q_t, double *discharge, int64_t out_rows,
                           int64_t num_output_steps, int64_t num_routing_per_output, void *stream);
int rr_unit_route_dev(rr_plan *plan, double *q_ch, double *q_full, const double *convolved_lateral,
                      int64_t conv_rows, double *discharge, int64_t out_rows,
                      int64_t num_runoff_steps, int64_t num_substeps, void *stream);
int rr_uh_convolve_dev(int device, const double *kernel, double *state, const double *lateral, double *out,
                       int64_t T, int64_t n_ks, int64_t n, void *stream);

/* The same with the routers' post-processing (river_route/routers/TransformMuskingum.py:128-142) fused into the pass that
 * returns the rows to params order: discharge32[o, i] = (float) mean_{j < factor} discharge[o * factor + j, i], (T / factor, n)
 * rows, 4 bytes written per value instead of 8 + 8 + 4.  RR_E_UNSUPPORTED when the call is not time-tiled or factor x
 * sub-steps does not divide 128: use the float64 form and rr_resample_cast_dev then. */
int rr_rapid_route_f32_dev(rr_plan *plan, double *q_t, const double *qlateral, int64_t ql_rows, float *discharge32,
                           int64_t num_runoff_steps, int64_t num_substeps, int64_t factor, void *stream);
/* rr_rapid_route_dev with float32 lateral rows, as qlateral files store them: 4 bytes read per value instead of 8, and the
 * routers upload half the bytes; float32 -> float64 is exact, so the result equals rr_rapid_route_dev on the converted rows
 * bit for bit.  Exactly one of discharge (float64 rows, out_rows >= 1 cyclic) and discharge32 (float32 rows, `factor` routed
 * rows averaged, as rr_rapid_route_f32_dev).  Time-tiled kernel only (RR_E_UNSUPPORTED otherwise: convert and call
 * rr_rapid_route_dev). */
int rr_rapid_route_f32in_dev(rr_plan *plan, double *q_t, const float *qlateral32, int64_t ql_rows, double *discharge, int64_t out_rows,
                             float *discharge32, int64_t factor, int64_t T, int64_t nsub, void *stream);
int rr_muskingum_route_f32_dev(rr_plan *plan, double *q_t, float *discharge32, int64_t num_output_steps,
                               int64_t num_routing_per_output, void *stream);
int rr_unit_route_f32_dev(rr_plan *plan, double *q_ch, double *q_full, const double *convolved_lateral, int64_t conv_rows,
                          float *discharge32, int64_t num_runoff_steps, int64_t num_substeps, int64_t factor, void *stream);

/* RapidMuskingum fed by gridded runoff (river_route/routers/TransformMuskingum.py:38-51 -> runoff.py:288-332 -> RapidMuskingum.py:
 * 19-33 in one call, one sub-step per row): the weights product of rr_runoff_to_qlateral_dev (same arguments, device arrays;
 * pass area: the routers route volumes) runs inside the pass that builds the engine's records, so the catchment inflow never
 * exists as (T, n) rows.  Exactly one of discharge (float64, T rows) / discharge32 (float32, T / factor rows) is non-NULL.
 * RR_E_UNSUPPORTED where the call is not time-tiled (use rr_runoff_to_qlateral_dev + rr_rapid_route_dev). */
int rr_rapid_route_runoff_dev(rr_plan *plan, double *q_t, int64_t n_points, const int32_t *indptr, const int32_t *indices,
                              const double *weights, const void *runoff, int runoff_is_f32, int64_t stride_t, int64_t stride_p,
                              const double *area, int flags, double *discharge, float *discharge32, int64_t factor,
                              int64_t num_runoff_steps, void *stream);

/* UnitMuskingum with the unit-hydrograph convolution fused into the pass that turns rows into the engine's records
 * (river_route/routers/UnitMuskingum.py:72-98 in one call): depth[T*n] runoff depths, uh_kernel[n_ks*n], uh_state[n_ks*n]
 * in/out (carry-over, updated in place), n_ks <= 64.  Exactly one of discharge (float64, T rows) / discharge32 (float32,
 * T / factor rows, `factor` rows averaged) is non-NULL.  q_final[n] (may be NULL) receives the state the router keeps: the
 * last lateral inflow on headwaters, q_full on inner reaches.  The convolved lateral never exists as rows in memory.
 * RR_E_UNSUPPORTED where the call is not time-tiled (use rr_uh_convolve_dev + rr_unit_route_dev). */
int rr_unit_route_uh_dev(rr_plan *plan, double *q_ch, double *q_full, double *q_final, const double *uh_kernel, double *uh_state,
                         int64_t n_ks, const double *depth, double *discharge, float *discharge32, int64_t factor,
                         int64_t num_runoff_steps, int64_t num_substeps, void *stream);
/* The same with float32 runoff depths, as runoff files usually store them (4 bytes read and uploaded per value instead of 8;
 * float32 -> float64 is exact, so the results are those of the float64 copy, bit for bit). */
int rr_unit_route_uh_f32in_dev(rr_plan *plan, double *q_ch, double *q_full, double *q_final, const double *uh_kernel, double *uh_state,
                               int64_t n_ks, const float *depth32, double *discharge, float *discharge32, int64_t factor,
                               int64_t num_runoff_steps, int64_t num_substeps, void *stream);

/* ---- partitioned networks (multi-GPU): boundary reaches and streaming calls ----
 *
 * A network cut into parts (rr_partition_forest) is routed one part per GPU.  In the part that holds the
 * DOWNSTREAM end of a cut edge the upstream reach appears as a GHOST: a headwater column of the part's local
 * network whose discharge at every sub-step is prescribed from `ghost_series`; in the part that owns it the
 * reach is an EXPORT reach whose discharge after every sub-step is recorded in `export_series`.  Both series
 * are device arrays of shape (T * nsub, n_ghost) / (T * nsub, n_export), row = sub-step.  The streaming
 * calls keep the lag pipeline full while series arrive in batches (no drain between batches). */

/* Reaches are LOCAL params indices of this plan.  A ghost's value is prescribed, so whatever the local network puts upstream
 * of it is ignored.  UnitMuskingum distinguishes headwater tributaries from the others (_numba_kernels.py:150-156): a ghost
 * that mirrors a reach WITH upstream reaches must itself have one in the local network (a dummy headwater whose lateral
 * column is zero does), so that it falls on the same side of that distinction as the reach it mirrors, and it then has an
 * entry in the q_ch / q_full arrays, whose q_full is the mirrored reach's. */
int rr_plan_set_boundary(rr_plan *plan, int64_t n_ghost, const int64_t *ghost_reaches, int64_t n_export,
                         const int64_t *export_reaches);

/* Opens a routing call on device arrays (has_lateral = 1: RapidMuskingum, 0: channel-only Muskingum).
 * q_t[n]: initial state (ghost entries = the upstream reach's initial discharge).
 * lateral / discharge are read / written cyclically (step t: row t % lat_rows, row t % out_rows).  A caller that REFILLS a lateral
 * ring shorter than the call between two rr_stream_advance calls may announce at most lat_rows rows beyond those the engine has
 * taken; the engine takes them in whole batches -- 128 tick-rows (+ 15 of overlap) on the record path, K rows on the direct row path,
 * where K (rr_plan_reserve's info[1]) is capped to the largest multiple of 16 that fits BOTH rings (rings of fewer than 32 rows keep
 * to records) -- so a refilled ring must hold at least one such batch.  On the direct row path the columns of skeleton reaches are
 * patched into a discharge row info[6] (the pipeline's depth) rows after the row was first written: a cyclic discharge ring that the
 * caller drains while the call is open must be that long (or T rows), or the call must be reserved with RR_ROWS_NOT_PLAIN. */
int rr_stream_begin(rr_plan *plan, int has_lateral, const double *q_t, const double *lateral, int64_t lat_rows,
                    double *discharge, int64_t out_rows, int64_t T, int64_t nsub, const double *ghost_series,
                    double *export_series, void *stream);
/* Enqueues every routing tick whose inputs are present: lateral rows [0, lateral_rows_ready) and ghost
 * sub-steps [0, ghost_substeps_ready).  *export_substeps_ready = leading sub-steps of export_series that are
 * final once the enqueued work has run. */
int rr_stream_advance(rr_plan *plan, int64_t lateral_rows_ready, int64_t ghost_substeps_ready,
                      int64_t *export_substeps_ready);
/* Closes the call (all T steps must have been routed) and writes the final state to q_t[n] (may be NULL). */
int rr_stream_end(rr_plan *plan, double *q_t);

/* The same for UnitMuskingum (river_route/routers/_numba_kernels.py:88-171 on one part of a cut network): lateral = the
 * convolved runoff depths of this part's columns (rr_uh_convolve_dev; ghost columns are not read), q_ch / q_full over the
 * local reaches that have upstream reaches, ascending local index.  Export series hold the discharge a reach publishes
 * (q_full; the lateral inflow on a headwater).  Advance with rr_stream_advance. */
int rr_stream_begin_unit(rr_plan *plan, const double *q_ch, const double *q_full, const double *lateral, int64_t lat_rows,
                         double *discharge, int64_t out_rows, int64_t T, int64_t nsub, const double *ghost_series,
                         double *export_series, void *stream);
/* Closes the call and writes the final q_ch / q_full (both may be NULL). */
int rr_stream_end_unit(rr_plan *plan, double *q_ch, double *q_full);

/* Cuts a forest into at most n_parts balanced parts whose part graph is acyclic (boundary discharge flows one way):
 * main stems + the tributaries joining them farthest upstream in the last part, the other subtrees spread over the
 * rest (part graph of depth two); chain-like networks fall back to a nested min-max cut.
 * part_of[n] receives the part of every reach, parts are numbered upstream-first; part_sizes[n_parts] (may be NULL)
 * their sizes.  Host-only, needs no GPU. */
int rr_partition_forest(int64_t n, const int32_t *csc_indptr, const int32_t *csc_indices, int32_t n_parts,
                        int32_t *part_of, int64_t *part_sizes);

/* Depth-first post-order of a river network given as one downstream ROW index per reach (-1 at outlets), rows in ANY order:
 * order[k] = the row that comes k-th.  Every reach follows the whole sub-basin of each of its tributaries, so every sub-basin is a run
 * of consecutive rows -- still a valid order for river_route/tools.py:103-104 (upstream before downstream), and the one in which
 * rr_rapid_route_dev / rr_stream_begin route straight from and to the caller's rows (rr_plan_direct_info).  A reach's tributaries
 * are visited small sub-basins first (at most 256 reaches: what one tile of the direct row path holds; the largest of them first), then
 * the large ones, the largest last -- so the reaches of a main stem end up side by side behind the small sub-basins that join it, and the
 * 8-byte stores that patch them into the output rows share lines; outlets in ascending row index.  RR_E_INVALID for an index out of
 * range or a cycle.
 * Host-only, needs no GPU. */
int rr_postorder(int64_t n, const int64_t *down_index, int64_t *order);

/* ---- router post-processing on the device (river_route/routers/TransformMuskingum.py:128-142) ----
 * out[o, i] = (float) mean_{j < factor} discharge[o * factor + j, i]; num_rows must be a multiple of factor.
 * Halves (or better) the bytes that leave the GPU: the reference's routers hand float32 to their writer. */
int rr_resample_cast_dev(int device, const double *discharge, int64_t num_rows, int64_t n, int64_t factor, float *out,
                         void *stream);

/* ---- gridded runoff -> catchment lateral inflow (river_route/runoff.py:288-330) ----
 * qlateral[t, r] = sum over the CSR row r of weights[k] * runoff(t, indices[k]), then, in the reference's order:
 * RR_RUNOFF_CUMULATIVE input -> incremental (row t minus row t - 1, row 0 kept), RR_RUNOFF_FORCE_POSITIVE -> clip at 0,
 * NaN -> 0 (unless RR_RUNOFF_KEEP_NAN: the host resamples irregular time steps before filling), and, when
 * area != NULL, times area[r] (volumes, the routers' as_volumes=True).  weights = proportion * unit conversion.
 * runoff element (t, p) is at runoff[t * stride_t + p * stride_p] (float when runoff_is_f32, else double): pass the
 * block point-major (stride_t = 1, stride_p >= T) for contiguous gathers -- with stride_p a multiple of 16 and the rows
 * allocated in full (n_points * stride_p elements, 16-byte aligned) they are read as 16-byte vectors -- or as read
 * from the file (stride_p = 1).
 * The non-uniform-time resampling of the reference (pandas, runoff.py:313-325) stays on the host.
 * rr_runoff_to_qlateral takes host arrays; the _dev form takes device arrays and only enqueues on `stream`. */
#define RR_RUNOFF_CUMULATIVE 1
#define RR_RUNOFF_FORCE_POSITIVE 2
#define RR_RUNOFF_KEEP_NAN 4
int rr_runoff_to_qlateral(int device, int64_t n_rivers, int64_t n_points, int64_t T, const int32_t *indptr,
                          const int32_t *indices, const double *weights, const void *runoff, int runoff_is_f32,
                          int64_t stride_t, int64_t stride_p, const double *area, int flags, double *qlateral);
int rr_runoff_to_qlateral_dev(int device, int64_t n_rivers, int64_t n_points, int64_t T, const int32_t *indptr,
                              const int32_t *indices, const double *weights, const void *runoff, int runoff_is_f32,
                              int64_t stride_t, int64_t stride_p, const double *area, int flags, double *qlateral,
                              void *stream);

/* ---- small device helpers so a host language needs no HIP binding of its own ---- */
int rr_dev_malloc(int device, int64_t bytes, void **out);
int rr_dev_free(int device, void *ptr);
int rr_dev_upload(int device, void *dst_dev, const void *src_host, int64_t bytes);
int rr_dev_download(int device, void *dst_host, const void *src_dev, int64_t bytes);
int rr_dev_synchronize(int device);
/* Measured device copy rate (GB/s, bytes read + bytes written) of a 16-byte-per-lane copy kernel over `bytes` bytes,
 * `reps` launches: the achievable HBM rate bench.py reports beside the nominal 8 TB/s. */
int rr_copy_bandwidth(int device, int64_t bytes, int reps, double *gbps);

/* Rows between a file and a device array, for the routers' qlateral / discharge files (river_route/routers/TransformMuskingum.py:30-36,
 * Muskingum.py:319-352): n_rows rows of row_bytes bytes that lie in `path` at file_offset, file_pitch bytes apart (a NetCDF-3 fixed variable:
 * file_pitch = row_bytes; a record variable: the record size), go to / come from device rows dev_pitch bytes apart through pinned staging
 * chunks (reader / writer threads beside the copy engine): the (time, river) block never exists as a host array.  Bytes move as they are:
 * a big-endian float32 variable is converted on the device (rr_plan_set_row_format).  stream (may be NULL): work already enqueued there
 * is finished before the rows move; the functions return when they have. */
int rr_rows_upload(int device, void *dst_dev, int64_t dst_pitch, const char *path, int64_t file_offset, int64_t file_pitch, int64_t row_bytes,
                   int64_t n_rows, void *stream);
int rr_rows_download(int device, const void *src_dev, int64_t src_pitch, const char *path, int64_t file_offset, int64_t file_pitch,
                     int64_t row_bytes, int64_t n_rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RR_HIP_H */
