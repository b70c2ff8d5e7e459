/*
 * mmssl_hip.h — C ABI of libmmssl_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * MMSSL hot path (modality-aware GCN message passing, per-modality projection,
 * L2-normalise, InfoNCE and BPR losses).
 *
 * The upstream reference (HKUDS/MMSSL) has NO native/FFI interface: the path sits behind
 * PyTorch op call sites. Each entry point below therefore names the reference CALL SITE
 * it replaces (paths relative to /root/reference/MMSSL/). A maintainer binds these with
 * ctypes from torch.autograd.Function wrappers — see INTEGRATION.md and
 * mmssl_amd/_lib.py.
 *
 * Conventions
 *   - return 0 on success; <0 = MMSSL_E_* (bad argument / unsupported); >0 = hipError_t.
 *   - nothing throws across the ABI; no torch / C++ types in signatures.
 *   - every dense buffer is CALLER-owned device memory, row-major contiguous fp32,
 *     16-byte aligned; indices are int32 (graphs) or int64 (batch indices, like the
 *     reference's python ints -> LongTensor). Only `mmssl_graph` is library-owned.
 *   - all compute entry points are asynchronous on the passed hipStream_t (`stream`,
 *     e.g. torch.cuda.current_stream().cuda_stream); they never allocate, never
 *     synchronise and are hipGraph-capturable. mmssl_graph_create/destroy are
 *     synchronous set-up calls.
 *   - "workspace" arguments are scratch the caller allocates once (size from the
 *     matching *_workspace_bytes call) and may reuse across calls on the same stream.
 */
#ifndef MMSSL_HIP_H
#define MMSSL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMSSL_ABI_VERSION 1

#define MMSSL_E_BADARG   (-1)   /* null pointer, negative size, malformed CSR ...   */
#define MMSSL_E_UNSUPP   (-2)   /* feature width d not in {32,64,128,256} etc.      */
#define MMSSL_E_WORKSPACE (-3)  /* workspace too small                               */

int mmssl_abi_version(void);
/* Human-readable text for a return code of this library (static storage). */
const char* mmssl_strerror(int code);

/* ------------------------------------------------------------------------------------
 * Graph plan  — replaces the torch sparse COO handles the reference builds on the host
 *   Trainer.csr_norm -> matrix_to_tensor / sparse_mx_to_torch_sparse_tensor
 *   (main.py:89-112, 513-520; rebuilt per batch at main.py:378-397).
 * Input is the HOST CSR the reference's scipy code produces (values already normalised).
 * The plan holds, on the device of the current HIP context: the CSR (edges packed as
 * {col,val} pairs), its transpose (for backward: torch autograd's A^T . gradY), and a
 * degree-balanced work list (short rows -> one 16-lane group each, long rows split into
 * <=128-nnz wave slices, four slices of a row per block summed through LDS, rows beyond 512
 * edges combined across blocks in a fixed order).
 * nnz == 0 is legal (the reference's "empty modal graph" state).
 * ---------------------------------------------------------------------------------- */
typedef struct mmssl_graph mmssl_graph;

int mmssl_graph_create(const int32_t* rowptr, const int32_t* col, const float* val,
                       int32_t rows, int32_t cols, int64_t nnz, void* stream,
                       mmssl_graph** out);
/* The same with the XCD banding of the work list stated explicitly. xcd_bands: 0 = automatic (what mmssl_graph_create
 * does: a direction's short rows are banded when >= 50 % of its edges fall into their row's dominant column band and the
 * bands are balanced - a graph with community / locality structure whose rows and columns are numbered accordingly),
 * 1 = always, -1 = never. Banding
 * changes which block processes which row, never the arithmetic: results are bit-identical either way. */
int mmssl_graph_create_ex(const int32_t* rowptr, const int32_t* col, const float* val,
                          int32_t rows, int32_t cols, int64_t nnz, int xcd_bands, void* stream,
                          mmssl_graph** out);
/* ... and with the band of every row (row_band[rows]: which XCD's blocks process the row in Y = A.X) and of every column
 * (col_band[cols]: the same for the transposed product) GIVEN, entries in [0, 8): for graphs whose communities are not
 * contiguous in the numbering - the caller clusters rows and columns together (mmssl_amd/graph.py co-clusters them at plan
 * time) so that a row mostly references columns of its own band. Both NULL = mmssl_graph_create_ex. */
int mmssl_graph_create_banded(const int32_t* rowptr, const int32_t* col, const float* val,
                              int32_t rows, int32_t cols, int64_t nnz, int xcd_bands, const int32_t* row_band,
                              const int32_t* col_band, void* stream, mmssl_graph** out);
int mmssl_graph_destroy(mmssl_graph* g);
/* info[0..7] = rows, cols, nnz, group_items, wave_items, multi_rows, partial_slots,
 *              same four for the transpose in info[8..11]; info[12..14] = work-list shaping constants;
 *              info[15] = XCD banding: bit 0 / 1 = forward / transposed direction banded, bits 8..23 / 24..39 = the
 *              directions' locality scores in 1/1000. */
int mmssl_graph_info(const mmssl_graph* g, int64_t info[16]);
/* Copy the device-resident transposed CSR back to host buffers (tests / debugging). */
int mmssl_graph_export_transpose(const mmssl_graph* g, int32_t* t_rowptr, int32_t* t_col,
                                 float* t_val, void* stream);
/* (rowptr [rows+1], col, val) of A (transpose = 0) or A^T (1) to the host; col / val hold `cap` entries;
 * *nnz_out = stored entries (read from the device: works for device-built plans too). Synchronises. */
int mmssl_graph_export_f32(const mmssl_graph* g, int transpose, int32_t* rowptr, int32_t* col, float* val,
                           int64_t cap, int64_t* nnz_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Modal-graph rebuild on the device (main.py:378-405): both plans of the rebuild
 *   A_ui = csr_norm(csr_matrix(ones, (users, items)), mean_flag=True),  A_iu = csr_norm(its transpose, True)
 * from one (user, item) pair list that is already on the device (the batch users tiled k times and their top-k
 * items), with no host round trip, no allocation and no synchronisation: capturable in a hipGraph. A pair handle is
 * created once for a capacity (<= 16384 pairs) and rebuilt in place; mmssl_graph_pair_get returns the two graph
 * handles (owned by the pair; usable with every mmssl_spmm / mmssl_graph_* call). Duplicate pairs stay separate
 * edges of weight 1/sqrt(deg(row)) (= the reference's summed duplicates). n = 0 gives empty graphs.
 * ---------------------------------------------------------------------------------- */
typedef struct mmssl_graph_pair mmssl_graph_pair;
int mmssl_graph_pair_create(int32_t n_users, int32_t n_items, int64_t capacity, mmssl_graph_pair** out);
int mmssl_graph_pair_destroy(mmssl_graph_pair* h);
int mmssl_graph_pair_get(mmssl_graph_pair* h, mmssl_graph** ui, mmssl_graph** iu);
int mmssl_graph_pair_rebuild(mmssl_graph_pair* h, const int64_t* users, const int64_t* items, int64_t n,
                             void* stream);

/* Host-only planning helpers (pure CPU, no device needed) — exported so the host logic
 * is testable without a GPU; mmssl_graph_create uses exactly these. */
int mmssl_csr_validate_host(const int32_t* rowptr, const int32_t* col, int32_t rows,
                            int32_t cols, int64_t nnz);
int mmssl_csr_transpose_host(const int32_t* rowptr, const int32_t* col, const float* val,
                             int32_t rows, int32_t cols, int64_t nnz, int32_t* t_rowptr,
                             int32_t* t_col, float* t_val);
/* XCD banding of a plan's group items (no reference counterpart: an MI355X placement matter). The columns are cut into
 * n_bands equal ranges; band_of_row[r] = the band most of row r's edges fall into, *score = the fraction of all edges that
 * fall into their row's band (1 / n_bands for uniformly random columns). mmssl_plan_band_group_items_host reorders the
 * degree-sorted group items band-major (stable) and returns the band boundaries. */
int mmssl_plan_band_host(const int32_t* rowptr, const int32_t* col, int32_t rows, int32_t cols, int32_t n_bands,
                         int32_t* band_of_row, double* score);
int mmssl_plan_band_group_items_host(int32_t* group_items, int64_t n_g, const int32_t* band_of_row, int32_t n_bands,
                                     int32_t* band_start);
/* The wave items' side of the banding: the light section is reordered band-major in place and wmap[b] (ceil(n_w / 4)
 * entries, a permutation) names the wave block - 4 consecutive wave items - that hardware block b processes: one of band
 * b % n_bands while that band has any, heaviest first. */
int mmssl_plan_band_wave_blocks_host(int32_t* wave_items, int64_t n_w, const int32_t* band_of_row, int32_t n_bands,
                                     int32_t* wmap);
/* counts[0..3] = group_items, wave_items (incl. padding), multi_rows, partial_slots */
int mmssl_plan_count_host(const int32_t* rowptr, int32_t rows, int64_t counts[4]);
/* items are int32 quadruples {row, edge_begin, edge_end, code}:
 *   group items : rows with <= 32 edges (one lane group each), longest first, code -1;
 *   wave items  : <= 128-edge slices, [heavy section | light section]:
 *       light, code -1 : a whole row of 33..128 edges;
 *       heavy : rows with > 128 edges, heaviest first, each row padded with no-work items
 *               {-1,0,0,code} to a multiple of 4 so that one 4-wave block holds slices of ONE row
 *               (summed through LDS): code -2 = the row fits the block (<= 512 edges);
 *               code s >= 0 = the row spans several blocks and this block owns partial slot s;
 *   multi are quadruples {row, first_slot, n_slots, 0} for the rows that span several blocks. */
int mmssl_plan_fill_host(const int32_t* rowptr, int32_t rows, int32_t* group_items,
                         int32_t* wave_items, int32_t* multi);

/* ------------------------------------------------------------------------------------
 * SpMM  Y[R,d] = op(A) . X[C,d]   — replaces torch.sparse.mm / torch.mm(sparse, dense)
 *   MMSSL.mm (Models.py:69-73) call sites Models.py:177-186 and the GCN propagation
 *   Models.py:201-211; transpose=1 is the autograd backward gradX = A^T . gradY.
 * epilogue: MMSSL_EPI_NONE, or MMSSL_EPI_SOFTMAX = row softmax over the d features fused
 *   into the store (the last GCN layer, Models.py:202-204).
 * d in {32, 64, 128, 256}.
 * ---------------------------------------------------------------------------------- */
#define MMSSL_EPI_NONE    0
#define MMSSL_EPI_SOFTMAX 1
/* extended (mmssl_spmm_ex_f32 only): the two epilogues that let the whole backward of the GCN
 * chain (Models.py:199-214: layer mean + last-layer softmax) consist of SpMM launches alone */
#define MMSSL_EPI_AXPY             2  /* Y = A.X + alpha * Z[row]                               */
#define MMSSL_EPI_AXPY_SOFTMAX_BWD 3  /* t = A.X + alpha * Z[row]; Y = S[row] * (t - <t, S[row]>)  (softmax bwd) */
#define MMSSL_EPI_MASK             4  /* Y = keep ? A.X * scale : 0  (mmssl_spmm_mask_f32 only)                  */

/* ---- batch rows of the interaction pattern (SURVEY.md 8f "next #1") --------------------------------
 * The reference builds `torch.tensor(self.ui_graph_raw[users].todense()).cuda()` — a dense
 * [B, n_items] host matrix + upload — in every Trainer.u_sim_calculation call and for the
 * discriminator's real-data rows (main.py:281-298, 349). These read the same pattern from the plan's
 * device CSR (row u of g = the items of user u). rows: int64 device array of n row ids of g;
 * width must equal the number of columns of g; P/S/gS/gP/out are [n, width] fp32, contiguous.
 *   mask_normalize      in place: P[b, seen] = 0; P[b,:] /= max(|P[b,:]|, eps); inv_norm[b] = the factor
 *                       (u_sim = F.normalize(sim * (1 - R[users])), main.py:293-297)
 *   mask_normalize_bwd  gP = (1-R) * inv * (gS - S (S.gS))   (gS/eps for clamped rows)
 *   rows_dense          out[b,:] = value * R[rows[b],:] */
int mmssl_graph_rows_mask_normalize_f32(const mmssl_graph* g, const int64_t* rows, int64_t n, float* P,
                                        int64_t width, float eps, float* inv_norm, void* stream);
int mmssl_graph_rows_mask_normalize_bwd_f32(const mmssl_graph* g, const int64_t* rows, int64_t n,
                                            const float* S, const float* gS, const float* inv_norm,
                                            int64_t width, float eps, float* gP, void* stream);
int mmssl_graph_rows_dense_f32(const mmssl_graph* g, const int64_t* rows, int64_t n, float value,
                               float* out, int64_t width, void* stream);

/* ------------------------------------------------------------------------------------
 * Batch similarity rows and per-row top-K (csrc/simtopk.hip)
 *   Trainer.u_sim_calculation     main.py:283-298   (scores . (1 - R[users]), then F.normalize(dim=1))
 *   evaluation scoring + ranking  utility/batch_test.py:21-36, 91-100, 150-152
 * mmssl_sim_rows_f32: out[b, j] = < Q[qidx[b], :], T[j, :] > (qidx NULL: row b itself), fp32 MFMA tiles, d in
 *   {32, 64, 128, 256}; entries (b, c) with c in the CSR row qidx[b] of (mask_rowptr, mask_cols: int32, sorted per row;
 *   both NULL = no mask) become mask_value (0 for u_sim, -inf for the evaluation). out has row pitch ldo >= n_items.
 *   sumsq_part (may be NULL): [B, mmssl_sim_rows_parts(n_items)] partial sums of squares of the unmasked scores;
 *   mmssl_rows_scale_parts_f32 turns them into the row factors 1/max(norm, eps), applies them in place and
 *   returns them (inv_out may be NULL). mmssl_graph_sim_rows_f32: the same with a graph plan's CSR as the mask.
 * mmssl_usim_rows_f32 / mmssl_graph_usim_rows_f32: u_sim in ONE pass over the [B, n_items] matrix (d in {32, 64, 128}):
 *   out = F.normalize(scores with the masked entries at 0, dim = 1) and inv_out[b] = 1 / max(|row b|, eps). The row norms
 *   are known before the tile kernel runs - |S_b|^2 = q_b^T (T^T T) q_b - sum over the row's masked items of (q_b . t_j)^2,
 *   Gram matrix on the fp32 matrix pipe with float64 block sums, quadratic form in float64 - so the matrix is written
 *   once, already scaled; columns [n_items, ldo) of a pitched row are written as zeros. workspace:
 *   mmssl_usim_workspace_bytes(d, n_items) bytes (0 = d not supported: use the two-launch form above).
 * mmssl_topk_rows_f32: idx_out[b, 0..K) = columns of the K largest entries of row b in DESCENDING value, ties by
 *   ASCENDING column (heapq.nlargest over an ascending-id dict); K <= 256, n_cols <= 36864 per launch (wider rows:
 *   one launch per column block, then one over the blocks' winners - ops.topk_rows does that); rows shorter than K
 *   are padded with -1. val_out (may be NULL) receives the values.
 * mmssl_rows_membership_u8: out[b, k] = 1 iff cand[b, k] is a column of CSR row rows[b] (sorted columns): the
 *   hit matrix of the evaluation without a dense [users, items] positives matrix.
 * mmssl_eval_accumulate_f64: the metric formulas of utility/batch_test.py:38-80 / utility/metrics.py on the device:
 *   acc[m * 8 + i] += sum over the B users of metric m (0 precision, 1 recall, 2 ndcg, 3 hit ratio) @ ks[i], float64,
 *   from the ranked candidates cand [B, K] (the top-K kernel's output) and the users' positives (CSR rows rows[b], sorted
 *   int32 columns). Fixed summation order (threads, waves, blocks). acc is [4][8] doubles, zeroed by the caller before
 *   the first batch and read once after the last; n_ks <= 8, ks[i] <= K. workspace: mmssl_eval_workspace_bytes(B).
 * ---------------------------------------------------------------------------------- */
size_t mmssl_eval_workspace_bytes(int64_t B);
int mmssl_eval_accumulate_f64(const int32_t* pos_rowptr, const int32_t* pos_cols, const int64_t* rows, int64_t B, int K,
                              const int64_t* cand, const int* ks, int n_ks, double* acc, void* workspace,
                              size_t workspace_bytes, void* stream);
int mmssl_sim_rows_parts(int64_t n_items);
int mmssl_sim_rows_f32(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t n_items, int d,
                       const int32_t* mask_rowptr, const int32_t* mask_cols, float mask_value, float* out,
                       int64_t ldo, float* sumsq_part, void* stream);
size_t mmssl_usim_workspace_bytes(int d, int64_t n_items);
int mmssl_usim_rows_f32(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t n_items, int d,
                        const int32_t* mask_rowptr, const int32_t* mask_cols, float eps, float* out, int64_t ldo,
                        float* inv_out, void* workspace, size_t workspace_bytes, void* stream);
int mmssl_graph_usim_rows_f32(const mmssl_graph* g, const float* Q, const int64_t* rows, int64_t n, const float* T, int d,
                              float eps, float* out, int64_t ldo, float* inv_out, void* workspace, size_t workspace_bytes,
                              void* stream);
int mmssl_graph_sim_rows_f32(const mmssl_graph* g, const float* Q, const int64_t* rows, int64_t n, const float* T,
                             int d, float mask_value, float* out, int64_t ldo, float* sumsq_part, void* stream);
int mmssl_rows_scale_parts_f32(float* X, int64_t B, int64_t n_items, int64_t ldo, const float* sumsq_part,
                               int nparts, float eps, float* inv_out, void* stream);
/* mmssl_graph_rows_mask_normalize_bwd_f32 with separate row pitches for S, gS and gP */
int mmssl_graph_rows_mask_normalize_bwd_ld_f32(const mmssl_graph* g, const int64_t* rows, int64_t n, const float* S,
                                               int64_t ld_s, const float* gS, int64_t ld_g, const float* inv_norm,
                                               int64_t width, float eps, float* gP, int64_t ld_p, void* stream);
int mmssl_topk_rows_f32(const float* X, int64_t B, int64_t n_cols, int64_t ldx, int K, int64_t* idx_out,
                        float* val_out, void* stream);
int mmssl_rows_membership_u8(const int32_t* rowptr, const int32_t* cols, const int64_t* rows, int64_t B, int K,
                             const int64_t* cand, uint8_t* out, void* stream);

/* t = A.X + alpha * Z[row]; Y = S[row]*(t - <t,S[row]>)  */

/* Workspace of one launch = partial sums of the rows that span several blocks + their arrival counters.
 * It must be ZERO-FILLED ONCE by the caller before its first use (every launch leaves the counters zero
 * again). Launches that may overlap on different streams need different workspaces. */
size_t mmssl_spmm_workspace_bytes(const mmssl_graph* g, int transpose, int d);
int mmssl_spmm_f32(const mmssl_graph* g, int transpose, const float* X, int d, float* Y,
                   int epilogue, void* workspace, size_t workspace_bytes, void* stream);
/* Z, S: [rows_out, d] fp32 (NULL unless the epilogue reads them). */
int mmssl_spmm_ex_f32(const mmssl_graph* g, int transpose, const float* X, int d, float* Y,
                      int epilogue, const float* Z, float alpha, const float* S, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same product on COLUMN CHUNKS of wider row-major tables (no reference counterpart: the row-sharded step,
 * mmssl_amd/dist.py, propagates a d-wide table as d / chunk independent column chunks so that chunk c's product runs
 * under chunk c+1's RCCL collective - LightGCN propagation, Models.py:201-211, is independent per column):
 * X row j starts at X + j * ldx, Y / Z row i at Y + i * ldy, Z + i * ldy (floats; multiples of 4, >= d; base pointers
 * 16-byte aligned); d = the chunk's width. epilogue: MMSSL_EPI_NONE or MMSSL_EPI_AXPY (Y = op(A).X + alpha * Z). */
int mmssl_spmm_ld_f32(const mmssl_graph* g, int transpose, const float* X, int64_t ldx, int d, float* Y, int64_t ldy,
                      int epilogue, const float* Z, float alpha, void* workspace, size_t workspace_bytes,
                      void* stream);
/* Y = keep ? (op(A).X) * scale : 0 — the dropout backward of the modality projection (nn.Dropout, Models.py:54,
 * 173-174) fused into the SpMM that produces the projection's output gradient (autograd of Models.py:177, 182).
 * Y [R, d] packs d / dm modalities of dm features side by side; keep is the uint8 [d / dm, R, dm] mask layout of
 * mmssl_proj_fwd_f32. */
int mmssl_spmm_mask_f32(const mmssl_graph* g, int transpose, const float* X, int d, float* Y,
                        const uint8_t* keep, int dm, float scale, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * Row kernels
 *   mmssl_l2norm_rows*: F.normalize(x, p=2, dim=1) = x / max(||x||, eps)
 *     (Models.py:196-197, 217-218; main.py:212-213) and its backward.
 *     Y = alpha * normalize(X) + (Base ? Base : 0)   (alpha/Base fuse the "+ rate *"
 *     adds at Models.py:196-197,217-218).
 *   mmssl_softmax_rows_bwd: gX = Y * (gY - sum(gY*Y)) (backward of Models.py:203-204).
 *   mmssl_sumsq: sum of squares of a flat buffer -> out[0] (feat_reg main.py:252-257,
 *     BPR regulariser main.py:503); deterministic two-stage reduction.
 * ---------------------------------------------------------------------------------- */
int mmssl_l2norm_rows_f32(const float* X, const float* Base, float alpha, int64_t rows, int d,
                          float eps, float* Y, void* stream);
int mmssl_l2norm_rows_bwd_f32(const float* X, const float* gY, float alpha, int64_t rows, int d,
                              float eps, float* gX, void* stream);
int mmssl_softmax_rows_bwd_f32(const float* Y, const float* gY, float scale, int64_t rows, int d,
                               float* gX, void* stream);   /* gX = scale * Y*(gY - <gY,Y>) */
/* Y = softmax(X) over the d features of every row (torch.softmax(.., dim=-1), Models.py:203-204) as a launch of its own
 * - bit for bit the MMSSL_EPI_SOFTMAX store epilogue - for rows that are only complete after a reduce-scatter or after
 * all column chunks of a product have been written. Y == X allowed. */
int mmssl_softmax_rows_f32(const float* X, int64_t rows, int d, float* Y, void* stream);
/* Layer mean + modality fusion of the final embeddings (Models.py:213-218) in one pass:
 *   out = inv * sum_k layers[k] + r * normalize(A) + r * normalize(B)
 * `layers` is a HOST array of n_layers (<= 8) device pointers. If sumsq_part != NULL the kernel
 * also leaves per-block partials of sum(|A|^2 + |B|^2) there (mmssl_layer_combine_blocks() floats)
 * for the feature regulariser (main.py:252-257).
 * bwd: gA = r * normalize_bwd(A, G) + c * A,  gB likewise,  gL = inv * G  (gL / c may be NULL;
 * c = c_scale * c_dev[0], a device scalar: the incoming gradient of the sum of squares). */
int mmssl_layer_combine_blocks(int64_t rows, int d);
int mmssl_layer_combine_f32(const float* const* layers, int n_layers, float inv, const float* A,
                            const float* B, float r, int64_t rows, int d, float eps, float* out,
                            float* sumsq_part, void* stream);
int mmssl_layer_combine_bwd_f32(const float* A, const float* B, const float* G, float r, float inv,
                                const float* c_dev, float c_scale, int64_t rows, int d, float eps,
                                float* gA, float* gB, float* gL, void* stream);
/* Both sides of that backward (user tables: A0, B0, G0 ...; item tables: A1, B1, G1 ...) in ONE launch, each with the
 * arithmetic of mmssl_layer_combine_bwd_f32 (bitwise the same results); gL0 / gL1 may be NULL. */
int mmssl_layer_combine_bwd2_f32(const float* A0, const float* B0, const float* G0, int64_t rows0, float* gA0,
                                 float* gB0, float* gL0, const float* A1, const float* B1, const float* G1,
                                 int64_t rows1, float* gA1, float* gB1, float* gL1, float r, float inv,
                                 const float* c_dev, float c_scale, int d, float eps, void* stream);
/* out[0] = sum of `n` floats (fixed order, one block): second stage for the partials above. */
int mmssl_sum_partials_f32(const float* part, int64_t n, float* out, void* stream);
/* Layer mean + modality fusion over PACKED modal features (Models.py:213-218 for a modality list), up to two SIDES
 * (user tables, item tables) in one launch:
 *   out[k][row, :] = inv * sum_l layers[k][l][row, :] + r * sum_m normalize(Mod[k][row, m d : (m+1) d])
 * Mod[k] [rows[k], nm * d] holds the nm modal feature tables side by side (the output layout of the d * nm modal SpMM
 * chains); sumsq_part[k] (mmssl_fuse_blocks(rows[k], d, nm) entries; the array or an entry may be NULL) receives the
 * blocks' shares of sum |Mod[k]|^2 — the feature regulariser of main.py:252-257 — in a fixed order.
 * nm in {1, 2, 4}, d in {32, 64, 128, 256}, nm * d <= 256.
 *   backward: gMod[k][row, m-th slice] = r * normalize_bwd(Mod_m, G[k]) + (c_scale * c_dev[0]) * Mod_m
 *   (+ Gx[k][row, m-th slice] when given: gradients that arrive on the modal features themselves), gL[k] = inv * G[k]
 *   (the arrays Gx / gL or their entries may be NULL). */
int mmssl_fuse_blocks(int64_t rows, int d, int nm);
int mmssl_fuse_fwd_f32(int sides, const float* const* const* layers, int n_layers, float inv,
                       const float* const* Mod, int nm, float r, const int64_t* rows, int d, float eps,
                       float* const* out, float* const* sumsq_part, void* stream);
/* mmssl_fuse_fwd_f32's rows for a LIST of rows per side (idx[k]: n_idx[k] int64 row numbers, repeats allowed): only those
 * rows of out[k] are written, bit for bit the dense launch's values. A training step reads the fused tables at its batch
 * rows only, so the dense launch can leave its critical path: the regulariser's |Mod|^2 sums then fall out of
 * mmssl_fuse_bwd_f32 (sumsq_part[k]: mmssl_fuse_blocks(rows, d, nm) per-block partials of side k, may be NULL; it reads
 * every row of Mod for the norms anyway) and join the loss by mmssl_loss_add_partials_f32 (total[0] += c * sum(part);
 * sum_out, may be NULL, receives the sum). */
int mmssl_fuse_fwd_rows_f32(int sides, const float* const* const* layers, int n_layers, float inv,
                            const float* const* Mod, int nm, float r, const int64_t* const* idx, const int64_t* n_idx,
                            int d, float eps, float* const* out, void* stream);
/* The same on a ROW-SHARDED table (mmssl_amd/dist.py; no reference counterpart): idx[k] holds GLOBAL row ids, this rank owns
 * rows [lo[k], lo[k] + n_local[k]) of side k as local rows 0 .. n_local[k] - 1 and computes only the listed rows it owns
 * (the batch rows of the other ranks are theirs to compute; mmssl_gather_owned_rows_f32 zero-fills them). */
int mmssl_fuse_fwd_owned_rows_f32(int sides, const float* const* const* layers, int n_layers, float inv,
                                  const float* const* Mod, int nm, float r, const int64_t* const* idx,
                                  const int64_t* n_idx, const int64_t* lo, const int64_t* n_local, int d, float eps,
                                  float* const* out, void* stream);
int mmssl_loss_add_partials_f32(const float* part, int64_t n, float c, float* total, float* sum_out, void* stream);
int mmssl_fuse_bwd_f32(int sides, const float* const* Mod, int nm, const float* const* G, const float* const* Gx,
                       float r, float inv, const float* c_dev, float c_scale, const int64_t* rows, int d, float eps,
                       float* const* gMod, float* const* gL, float* const* sumsq_part, void* stream);
size_t mmssl_sumsq_workspace_bytes(int64_t n);
int mmssl_sumsq_f32(const float* X, int64_t n, float* out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Modality projection  Y[M,N] = dropout(F[M,K] . W[N,K]^T + b[N])
 *   nn.Linear image_trans / text_trans + nn.Dropout (Models.py:28-29, 54, 173-174).
 *   keep: optional uint8 [M,N] keep-mask (1 = keep); kept entries are scaled by `scale`
 *   (= 1/(1-p)); keep == NULL means no dropout (eval mode). fp32 MFMA
 *   (v_mfma_f32_32x32x2_f32), exact fp32. N % 4 == 0, K % 4 == 0; N <= 256 unless b == keep == NULL and
 *   K % 32 == 0.
 *   mmssl_linear_wgrad: gW[N,K] = gYm[M,N]^T . F[M,K], gb[N] = column sums of gYm, where
 *   gYm = gY * keep * scale is the dropout backward, applied while gY is fetched (keep == NULL:
 *   gYm = gY). Autograd of the same call site.
 * ---------------------------------------------------------------------------------- */
size_t mmssl_linear_workspace_bytes(int64_t M, int K, int N);   /* split-K partials */
int mmssl_linear_f32(const float* F, const float* W, const float* b, const uint8_t* keep,
                     float scale, int64_t M, int K, int N, float* Y, void* workspace,
                     size_t workspace_bytes, void* stream);
/* G [M, N] fp32 (optionally dropout-masked: keep/scale) -> T [N, Mp] fp32 = G^T, zero in columns M..Mp-1
 * (Mp % 4 == 0); colsum (may be NULL) receives the column sums of the masked G, i.e. the bias gradient.
 * With it the weight gradient of the projection (autograd of Models.py:173-174) is the FORWARD product
 *   gW [N, K] = mmssl_linear_f32(F = T [N, Mp], W = F^T [K, Mp], b = NULL, keep = NULL, M = N, K = Mp, N = K)
 * against a transposed copy of the constant feature matrix (mmssl_linear_f32 accepts N > 256 for that plain
 * product when K % 32 == 0). */
size_t mmssl_transpose_mask_workspace_bytes(int64_t Mp, int N);
int mmssl_transpose_mask_f32(const float* G, const uint8_t* keep, float scale, int64_t M, int N, int64_t Mp,
                             float* T, float* colsum, void* workspace, size_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------------------------
 * Grouped modality projection: ALL modality problems of one step in one stream-K launch (csrc/projection.hip).
 *   image_trans / text_trans (+ further modalities) + nn.Dropout and their autograd, Models.py:28-29, 54, 173-174,
 *   for the whole modality list at once. Every problem g has the same M (items) and N == 64 channels:
 *     forward  Y[m, 64 g + n] = dropout(F_g[M, K_g] . W_g[64, K_g]^T + b_g)      Y is [M, ldy], ldy >= 64 n_prob
 *              (the modalities side by side: the layout the d = 64 n_prob modal SpMM chains consume); K_g % 32 == 0.
 *              Dropout: `keep` = given uint8 masks [n_prob, M, 64] (1 = keep), OR `keep_out` + `rng_state` + p_drop:
 *              the masks are drawn in the epilogue with the generator of mmssl_dropout_mask_u8 (same bytes as ONE
 *              mmssl_dropout_mask_u8 launch over n_prob * M * 64 elements with the same state) and written to keep_out
 *              for the backward; the caller advances rng_state[1] (mmssl_dropout_mask_ex_u8's external-tick contract).
 *              keep == keep_out == NULL: no dropout. Kept entries are scaled by `scale`.
 *     wgrad    gW_g[64, K_g] = G[:, 64 g : 64 g + 64]^T . F_g,  gb_g[64] = column sums of that slice of G; G [M, ldg] is
 *              the ALREADY dropout-masked output gradient (mmssl_spmm_ex_f32's MMSSL_EPI_MASK epilogue); K_g % 4 == 0.
 *   mmssl_proj_supported: 1 if the shape runs here (else use mmssl_linear_f32 / mmssl_linear_wgrad_f32 per problem).
 *   fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32, deterministic (fixed-order partial sums).
 * ---------------------------------------------------------------------------------- */
#define MMSSL_PROJ_MAX_PROBLEMS 4
int mmssl_proj_supported(int n_prob, const int* K, int64_t M, int N, int wgrad);
size_t mmssl_proj_workspace_bytes(int n_prob, const int* K, int64_t M, int N, int wgrad);
int mmssl_proj_fwd_f32(int n_prob, const float* const* F, const float* const* W, const float* const* bias,
                       const int* K, int64_t M, int N, const uint8_t* keep, uint8_t* keep_out,
                       const uint64_t* rng_state, float p_drop, float scale, float* Y, int64_t ldy,
                       void* workspace, size_t workspace_bytes, void* stream);
int mmssl_proj_wgrad_f32(int n_prob, const float* G, int64_t ldg, const float* const* F, const int* K, int64_t M,
                         int N, float* const* gW, float* const* gb, void* workspace, size_t workspace_bytes,
                         void* stream);
/* The same, with the AdamW update of the projection weights and biases (mmssl_adamw_ex_f32's rule and step-counter
 * contract: state[0], external_tick -> pre_ticked) applied by the epilogue to the gradient it has just summed:
 * W_g, b_g and their moments are updated in place, gW / gb (arrays or entries may be NULL) still receive the gradients.
 * b / mb / vb may be NULL (no bias). The optimiser launch for these tensors disappears from the step's critical path. */
int mmssl_proj_wgrad_adamw_f32(int n_prob, const float* G, int64_t ldg, const float* const* F, const int* K, int64_t M,
                               int N, float* const* gW, float* const* gb, float* const* W, float* const* mW,
                               float* const* vW, float* const* b, float* const* mb, float* const* vb,
                               const float* state, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int pre_ticked, void* workspace, size_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------------------------
 * The same grouped projection in SPLIT PRECISION on the bf16 matrix pipe (csrc/projection.hip, "projx"): every fp32 operand
 * value is cut exactly into three bf16 pieces and a product is evaluated as its six partial products of weight >= 2^-16
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulation): per product the dropped terms are <= 2^-23 relative - one fp32 rounding -
 * so the results are fp32-accurate (tests pin the error against float64 at or below the fp32-MFMA kernels' and torch's fp32
 * GEMM), at 6/16 of the fp32-MFMA issue time: the launch is bound by the feature stream (HBM) instead of the matrix pipe.
 * Same call sites as above (Models.py:28-29, 54, 173-174), same epilogues, same determinism. Differences at the boundary:
 *   - the constant feature matrices (Models.py:46-47) are passed as TILE-MAJOR IMAGES made once by mmssl_projx_pack_f32:
 *     `Fimg[g]` = image of F_g [M, K_g] for the forward, `FTimg[g]` = image of F_g^T [K_g, M] (transpose = 1) for the weight
 *     gradient; an image holds mmssl_projx_image_floats(rows, red) floats (rows x red zero-padded to 256 x 32 blocks, each
 *     block contiguous and laid out as the kernel's LDS stage, so every LDS-DMA instruction streams one contiguous KB);
 *   - K_g % 4 == 0 both ways (zero padding replaces the forward's K_g % 32 requirement);
 *   - workspace: mmssl_projx_workspace_bytes (256-byte aligned pointer): partial slots + the per-launch bf16 operand planes
 *     of W (forward) or of G^T (weight gradient, which also yields the bias-gradient sums);
 *   - n_blocks: the launch's block count = equal ranges the work is cut into (0 = one per CU). The launch is bound by the
 *     feature stream, which fewer CUs still saturate: a caller that runs other kernels beside it (the step's GCN chain on a
 *     side stream) passes ~13/16 of the CU count so that those kernels are not starved. Same value for the size query.
 *     Results are bit-identical for a given n_blocks (fixed-order partial sums), not across different ones.
 * ---------------------------------------------------------------------------------- */
int mmssl_projx_supported(int n_prob, const int* K, int64_t M, int N);
size_t mmssl_projx_image_floats(int64_t rows, int64_t red);
int mmssl_projx_pack_f32(const float* F, int64_t M, int64_t K, int64_t ldf, int transpose, float* out, void* stream);
size_t mmssl_projx_workspace_bytes(int n_prob, const int* K, int64_t M, int N, int wgrad, int n_blocks);
/* The weights' bf16 planes as a caller-owned image (mmssl_projx_wimg_bytes bytes, 256-byte aligned): mmssl_projx_wsplit_f32
 * makes it from W (the launch mmssl_projx_fwd_f32 runs in front of its main kernel), mmssl_projx_fwd_img_f32 is the forward
 * on an image made earlier - a step makes it right behind the optimiser's update of the weights, i.e. at the END of the
 * previous step, and its next forward starts with the main kernel. The caller answers for the image being that of the
 * current weights (mmssl_amd/ops.py keys it on the tensors' versions). */
size_t mmssl_projx_wimg_bytes(int n_prob, const int* K);
int mmssl_projx_wsplit_f32(int n_prob, const float* const* W, const int* K, void* wimg, void* stream);
int mmssl_projx_fwd_img_f32(int n_prob, const float* const* Fimg, const void* wimg, const float* const* bias, const int* K,
                            int64_t M, int N, const uint8_t* keep, uint8_t* keep_out, const uint64_t* rng_state,
                            float p_drop, float scale, float* Y, int64_t ldy, int n_blocks, void* workspace,
                            size_t workspace_bytes, void* stream);
int mmssl_projx_fwd_f32(int n_prob, const float* const* Fimg, const float* const* W, const float* const* bias,
                        const int* K, int64_t M, int N, const uint8_t* keep, uint8_t* keep_out,
                        const uint64_t* rng_state, float p_drop, float scale, float* Y, int64_t ldy,
                        int n_blocks, void* workspace, size_t workspace_bytes, void* stream);
int mmssl_projx_wgrad_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg, const int* K, int64_t M,
                          int N, float* const* gW, float* const* gb, int n_blocks, void* workspace,
                          size_t workspace_bytes, void* stream);
int mmssl_projx_wgrad_adamw_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg, const int* K, int64_t M,
                                int N, float* const* gW, float* const* gb, float* const* W, float* const* mW,
                                float* const* vW, float* const* b, float* const* mb, float* const* vb,
                                const float* state, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int pre_ticked, int n_blocks, void* workspace, size_t workspace_bytes, void* stream);
/* the same, and - `wimg` != NULL: a mmssl_projx_wsplit_f32 image of W - the epilogue also rewrites the bf16 planes of every
 * weight it updates (the zero padding past K is left alone): the image stays that of the current weights, the next
 * mmssl_projx_fwd_img_f32 needs no split launch in front of it */
int mmssl_projx_wgrad_adamw_img_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg, const int* K, int64_t M,
                                    int N, float* const* gW, float* const* gb, float* const* W, float* const* mW,
                                    float* const* vW, float* const* b, float* const* mb, float* const* vb, const float* state,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int pre_ticked,
                                    void* wimg, int n_blocks, void* workspace, size_t workspace_bytes, void* stream);
size_t mmssl_linear_wgrad_workspace_bytes(int64_t M, int K, int N);
/* 1 when mmssl_linear_wgrad_f32 will run this shape on the register-direct kernel, which applies keep/scale and
 * sums the bias gradient on the fragments it loads (pass `keep`; no separate dropout-backward pass is needed);
 * 0 when it runs the register-staged kernel, for which a pre-masked gY (mmssl_mask_scale_f32) measured faster. */
int mmssl_linear_wgrad_fuses_mask(int64_t M, int K, int N);
int mmssl_linear_wgrad_f32(const float* gY, const uint8_t* keep, float scale, const float* F,
                           int64_t M, int K, int N, float* gW, float* gb, void* workspace,
                           size_t workspace_bytes, void* stream);
/* out = g * keep * scale over n (multiple of 4) elements: the dropout backward as one pass. */
/* The same for PACKED modal rows: G [rows, nm * dm] = nm modalities side by side, keep = the uint8 [nm, rows, dm] masks of
 * mmssl_proj_fwd_f32 (dm % 4 == 0); out may alias G. The row-sharded step applies it after the reduce-scatter of the
 * partial A^T products (the unsharded step uses mmssl_spmm_mask_f32's epilogue instead). */
int mmssl_mask_packed_f32(const float* G, const uint8_t* keep, float scale, int64_t rows, int nm, int dm, float* out,
                          void* stream);
int mmssl_mask_scale_f32(const float* g, const uint8_t* keep, float scale, int64_t n, float* out,
                         void* stream);
/* Dropout keep-mask (nn.Dropout(p), Models.py:54): keep[i] = 1 with probability 1-p, from
 * Philox4x32-10 keyed by rng_state[0] (seed) and counted by (i/4, rng_state[1]); the launch advances
 * rng_state[1] itself, so hipGraph replays draw fresh masks. rng_state: 3 x uint64 in device memory
 * {seed, launch counter, 0}; n multiple of 4. The stream of masks is this library's own (the
 * reference's depends on torch's CUDA generator and is not reproducible across devices either). */
int mmssl_dropout_mask_u8(uint64_t* rng_state, float p, int64_t n, uint8_t* keep, void* stream);

/* AdamW update of up to MMSSL_ADAMW_MAX_TENSORS fp32 tensors in ONE launch — torch.optim.AdamW as
 * built at main.py:76-80 (amsgrad=False, maximize=False):
 *   p *= 1 - lr*wd; m += (g-m)(1-b1); v = v*b2 + (1-b2) g*g;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   t = state[0] + 1
 * params/grads/exp_avg/exp_avg_sq/numel are HOST arrays of `count` device pointers / sizes (16-B
 * aligned); state = 2 floats in device memory {completed steps, 0}: the launch increments
 * state[0] itself (graph-replay safe). */
#define MMSSL_ADAMW_MAX_TENSORS 24
int mmssl_adamw_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, int count, float* state, float lr,
                    float beta1, float beta2, float eps, float weight_decay, void* stream);

/* total[0] = sum_k w[k] * terms[k] (k < n <= 16) + c * extra[0]: the scalar loss assembly of
 * main.py:420 in one launch (terms / w / extra are device arrays; extra may be NULL). */
int mmssl_loss_assemble_f32(const float* terms, const float* w, int n, const float* extra, float c,
                            float* total, void* stream);
/* Batch rows of a row-sharded table (no reference counterpart: the reference is single-GPU):
 * out[j, :] = table[idx[j] - lo, :] if lo <= idx[j] < lo + rows_local else 0 (d % 4 == 0), and its adjoint
 * gtable[idx[j] - lo, :] += g[j, :] for the rows this rank owns (gtable pre-zeroed; fp32 atomics). */
int mmssl_gather_owned_rows_f32(const float* table, int64_t rows_local, int d, const int64_t* idx, int64_t n,
                                int64_t lo, float* out, void* stream);
int mmssl_scatter_owned_rows_f32(const float* g, const int64_t* idx, int64_t n, int64_t lo, int64_t rows_local,
                                 int d, float* gtable, void* stream);
/* The same launch also advances up to 4 float and 4 uint64 device counters by one (the AdamW step counters and
 * the dropout launch counter of a whole captured step: see the *_ex entry points, external_tick = 1). */
int mmssl_loss_assemble_tick_f32(const float* terms, const float* w, int n, const float* extra, float c,
                                 float* total, float* const* f32_ticks, int n_f32,
                                 uint64_t* const* u64_ticks, int n_u64, void* stream);
/* external_tick = 1: the launch does NOT advance its counter; a stream-ordered launch between the dropout and the
 * optimiser does (mmssl_loss_assemble_tick_f32). For AdamW the counter then already holds the number of THIS step
 * when the update runs; for the dropout mask it is advanced after use, as before. */
int mmssl_adamw_ex_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, int count, float* state, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int external_tick, void* stream);
/* The same update with SLICED gradients: tensor t's gradient is grads[t][i] + grads[t][gstride[t] + i] + ... over
 * slices[t] slices, added in slice order - split partials a producer leaves behind, consumed without a reduce launch
 * (bitwise the result of reducing first). slices == gstride == NULL is mmssl_adamw_ex_f32. */
int mmssl_adamw_sliced_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const int64_t* numel, const int32_t* slices,
                           const int64_t* gstride, int n_tensors, float* state, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int external_tick, void* stream);
int mmssl_dropout_mask_ex_u8(uint64_t* rng_state, float p, int64_t n, uint8_t* keep, int external_tick,
                             void* stream);
/* counter[0] += 1 on the stream: advances a generator's launch counter (rng_state + 1) when the masks were drawn inside
 * another kernel (mmssl_proj_fwd_f32's epilogue) or with external_tick set. */
int mmssl_tick_u64(uint64_t* counter, void* stream);
/* dst[0 .. count) = ring[(step_counter[0] % n_slots) * count ...]: a captured step reads its batch indices
 * (Data.sample() output, load_data.py:153-191, uploaded ahead of time) from a device-resident ring by a uint64 count of
 * completed steps that the step itself advances (mmssl_tick_u64, or the u64 tick list of its loss tail) - no host-side
 * copy between two replays. (Not the fp32 AdamW step counter: that one saturates at 2^24 steps.) */
int mmssl_select_slot_i64(const int64_t* ring, int n_slots, int64_t count, const uint64_t* step_counter, int64_t* dst,
                          void* stream);
/* Backward of the above: gterms[k] = g[0] * w[k], gextra[0] = g[0] * c (gextra may be NULL). */
int mmssl_loss_assemble_bwd_f32(const float* g, const float* w, int n, float c, float* gterms,
                                float* gextra, void* stream);

/* ------------------------------------------------------------------------------------
 * InfoNCE  — Trainer.batched_contrastive_loss + Trainer.sim (main.py:211-249)
 *   loss = mean_i -log( e^{c12_ii/tau} / (sum_j e^{c11_ij/tau} + sum_j e^{c12_ij/tau}
 *                        - e^{c11_ii/tau}) + 1e-8 ),  c = cosine of L2-normalised rows.
 *   z1, z2: raw (un-normalised) fp32 rows. idx == NULL: both are [n, d] and row i pairs with
 *   row i (the reference's signature, called as f(table1[users], table2[users]), main.py:411-412).
 *   idx != NULL (int64 [n], device): z1/z2 are whole tables and batch row i is table row idx[i] —
 *   the gather is fused into the first kernel and the backward scatter-adds into the
 *   caller-zeroed table gradients. The reference's 1024-row blocking is mathematically the
 *   full-matrix formula; any n >= 1 is accepted.
 *   fwd writes loss[0] and keeps what bwd needs in `workspace`; bwd(gloss) -> gz1, gz2 (either
 *   may be NULL when that input needs no gradient).  d in {32, 64, 128, 256}.
 * ---------------------------------------------------------------------------------- */
size_t mmssl_infonce_workspace_bytes(int64_t n, int d);
int mmssl_infonce_fwd_f32(const float* z1, const float* z2, const int64_t* idx, int64_t n, int d,
                          float tau, float* loss, void* workspace, size_t workspace_bytes,
                          void* stream);
/* mmssl_infonce_fwd_f32 with a caller-chosen constant inside the logarithm (the trainer's variant uses 1e-8,
 * main.py:244; Models.batched_contrastive_loss, Models.py:79-98, and the MICRO baseline use 0). Backward:
 * mmssl_infonce_bwd_f32 on the same workspace. */
int mmssl_infonce_fwd_eps_f32(const float* z1, const float* z2, const int64_t* idx, int64_t n, int d, float tau,
                              float log_eps, float* loss, void* workspace, size_t workspace_bytes, void* stream);
int mmssl_infonce_bwd_f32(const int64_t* idx, int64_t n, int d, float tau, const float* gloss,
                          float* gz1, float* gz2, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Batched form: n_problems (<= 4) losses that SHARE z2 and differ in z1 — the reference evaluates the
 * loss once per modality against the same user embeddings (main.py:411-412). One set of launches for
 * all problems; z1s / gz1s are HOST arrays of device pointers, losses / gloss device arrays
 * [n_problems]; gz2 receives the sum over the problems. */
size_t mmssl_infonce_multi_workspace_bytes(int n_problems, int64_t n, int d);
int mmssl_infonce_multi_fwd_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                int n_problems, int64_t n, int d, float tau, float* losses,
                                void* workspace, size_t workspace_bytes, void* stream);
/* The forward in two phases: bit 0 = row terms (prep, pair tiles, per-row log terms: everything the backward
 * needs), bit 1 = the loss scalars from the row terms. A caller that already knows the upstream gradient of the
 * losses can start the backward right after phase 1 and leave phase 2 off its critical path. */
int mmssl_infonce_multi_fwd_phase_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                      int n_problems, int64_t n, int d, float tau, float* losses,
                                      void* workspace, size_t workspace_bytes, int phases, void* stream);
/* The whole forward without the separate loss-reduction launch: the last row-term block of each problem to arrive
 * reduces that problem's loss (same arithmetic, same value). tickets: n_problems ints, 0 on entry, left 0. */
int mmssl_infonce_multi_fwd_ticket_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                       int n_problems, int64_t n, int d, float tau, float* losses,
                                       void* workspace, size_t workspace_bytes, int* tickets, void* stream);
int mmssl_infonce_multi_bwd_f32(const int64_t* idx, int n_problems, int64_t n, int d, float tau,
                                const float* gloss, float* const* gz1s, float* gz2, void* workspace,
                                size_t workspace_bytes, void* stream);
/* The same in two separately launchable phases so that independent work (the BPR backward, which
 * scatters into the same table gradient and must stay ordered before phase 2) can overlap phase 1:
 * phases bit 0 = pair tiles (workspace only), bit 1 = finish (diagonal terms, normalise-backward,
 * scatter-add into gz1s / gz2). phases == 3 is mmssl_infonce_multi_bwd_f32. */
/* The hot step's loss chain with mmssl_bpr_step_f32 in TWO parts (d in {32, 64}; same arithmetic, same bits as
 * mmssl_bpr_step_f32): its ROWS part (gathers, scores, scatter-added gradients, per-block partials into bpr_workspace) rides as
 * guest blocks of the forward chain's short prep launch - mmssl_infonce_multi_fwd_ticket_bpr_f32 = mmssl_infonce_multi_
 * fwd_ticket_f32 + those guests - and its one-block ASSEMBLY part (BPR loss, total = w . terms + c * extra, counter ticks)
 * as a guest of the backward finish: mmssl_infonce_multi_bwd_finish_bpr_f32 = mmssl_infonce_multi_bwd_phase_f32(phase 2)
 * + that block. Between them the backward pair tiles run WITHOUT guests (mmssl_infonce_multi_bwd_phase_f32, phase 1):
 * 512 pair-tile blocks fill the chip exactly once at two blocks per CU, B/16 guests behind them open a second round
 * (measured: pair tiles 34.7 -> 29.4 us, prep 5.7 -> 8.4 us, the chain 3 us shorter). gEu / gEi must be zero-filled
 * before the first of the three calls.
 * ROW TERMS ARE DEFERRED in this chain: mmssl_infonce_multi_fwd_ticket_bpr_f32 stops after the forward pair tiles (it
 * does not write `losses`, `tickets` is unused), the backward pair tiles compute the per-row coefficients from the
 * partial denominators on the fly (the row-terms launch, 8 us in the step for a few hundred flops per row, is gone) and
 * mmssl_infonce_multi_bwd_finish_bpr_f32 writes the n_problems InfoNCE losses to `losses` - which may alias entries of
 * `terms`: they are written before the assembly reads them. The three calls belong together, in this order. */
int mmssl_infonce_multi_fwd_ticket_bpr_f32(const float* const* z1s, const float* z2, const int64_t* idx, int n_problems,
                                           int64_t n, int d, float tau, float* losses, void* workspace,
                                           size_t workspace_bytes, int* tickets, const float* Eu, const float* Ei,
                                           const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B,
                                           float decay, int64_t batch_size, const float* g_mf, const float* g_emb,
                                           float* gEu, float* gEi, void* bpr_workspace, size_t bpr_workspace_bytes,
                                           void* stream);
int mmssl_infonce_multi_bwd_finish_bpr_f32(const int64_t* idx, int n_problems, int64_t n, int d, float tau,
                                           const float* gloss, float* losses, float* const* gz1s, float* gz2, void* workspace,
                                           size_t workspace_bytes, const float* Eu, const float* Ei, const int64_t* users,
                                           const int64_t* pos, const int64_t* neg, int64_t B, float decay,
                                           int64_t batch_size, const float* g_mf, const float* g_emb, float* gEu,
                                           float* gEi, float* terms, const float* w, int n_terms, const float* extra,
                                           float c, float* total, float* const* f32_ticks, int n_f32,
                                           uint64_t* const* u64_ticks, int n_u64, void* bpr_workspace,
                                           size_t bpr_workspace_bytes, const float* extra_parts, int64_t n_extra_parts,
                                           void* stream);
int mmssl_infonce_multi_bwd_phase_f32(const int64_t* idx, int n_problems, int64_t n, int d, float tau,
                                      const float* gloss, float* const* gz1s, float* gz2,
                                      void* workspace, size_t workspace_bytes, int phases, void* stream);

/* ------------------------------------------------------------------------------------
 * BPR  — gathers (main.py:368-370) + Trainer.bpr_loss (main.py:499-511)
 *   out3 = { mf_loss = -mean logsigmoid(u.p - u.n),
 *            emb_loss = decay * 0.5*(|u|^2+|p|^2+|n|^2) / batch_size,
 *            reg_loss = 0 }
 *   Eu [U,d], Ei [I,d]; users/pos/neg: int64 [B] device indices, or all NULL for the
 *   already-gathered form (row b of Eu / Ei_pos / Ei_neg; then Ei = pos rows and
 *   `Ei_neg` = neg rows).  bwd accumulates (atomicAdd) g_mf*d(mf) + g_emb*d(emb) into
 *   gEu / gEi, which the caller zero-fills (dense table gradients, like autograd's
 *   index backward).
 * ---------------------------------------------------------------------------------- */
size_t mmssl_bpr_workspace_bytes(int64_t B);
int mmssl_bpr_fwd_f32(const float* Eu, const float* Ei, const float* Ei_neg, const int64_t* users,
                      const int64_t* pos, const int64_t* neg, int64_t B, int d, float decay,
                      int64_t batch_size, float* out3, void* workspace, size_t workspace_bytes,
                      void* stream);
int mmssl_bpr_bwd_f32(const float* Eu, const float* Ei, const float* Ei_neg, const int64_t* users,
                      const int64_t* pos, const int64_t* neg, int64_t B, int d, float decay,
                      int64_t batch_size, const float* g_mf, const float* g_emb, float* gEu,
                      float* gEi, float* gEi_neg, void* stream);
/* The hot step's loss tail as ONE launch (main.py:368-371, 420, 499-511 for a caller that knows the upstream
 * gradients): BPR backward of the gathered rows for the device scalars g_mf / g_emb (scatter-add into gEu / gEi, which
 * the caller zero-filled or already holds other gradients), the BPR loss values terms[0..2] = (mf, emb, 0) by a
 * last-arriving-block reduction, total = sum_k w[k] * terms[k] + c * extra[0] (terms[3..n_terms-1] are read), and +1 on
 * the given counters (see mmssl_loss_assemble_tick_f32). workspace: mmssl_bpr_workspace_bytes(B); ticket: one int,
 * 0 on entry, left 0. extra_parts != NULL: the extra term arrives as n_extra_parts partial sums (the forward's
 * regulariser partials, mmssl_layer_combine_f32); they are reduced here (the arithmetic of mmssl_sum_partials_f32) and
 * the sum is also STORED to extra[0], which then is an output. */
int mmssl_bpr_step_f32(const float* Eu, const float* Ei, const int64_t* users, const int64_t* pos,
                       const int64_t* neg, int64_t B, int d, float decay, int64_t batch_size,
                       const float* g_mf, const float* g_emb, float* gEu, float* gEi, float* terms,
                       const float* w, int n_terms, const float* extra, float c, float* total,
                       float* const* f32_ticks, int n_f32, uint64_t* const* u64_ticks, int n_u64,
                       void* workspace, size_t workspace_bytes, int* ticket, const float* extra_parts,
                       int64_t n_extra_parts, void* stream);

/* ------------------------------------------------------------------------------------
 * Baseline models next to MMSSL (csrc/baselines.hip; SURVEY.md 8f "next #4")
 *   item-graph product  LATTICE/codes/Models.py:103-104, MICRO/codes/Models.py:60-64, 112-118: h' = item_adj . h with the
 *   kNN item graph kept as LISTS idx [rows, k] (int64 neighbour ids) / w [rows, k] instead of a dense or COO N x N matrix:
 *     mmssl_ell_spmm_f32      Y[i] = sum_j w[i, j] * H[idx[i, j]]                      (list order: deterministic)
 *     mmssl_ell_spmm_bwd_f32  gW[i, j] = < gY[i], H[idx[i, j]] > (the learned graph's gradient; may be NULL),
 *                             gH[idx[i, j]] += w[i, j] * gY[i] (fp32 atomics into the caller-ZEROED gH; may be NULL)
 *   NGCF layer  LATTICE/codes/Models.py:106-118, MICRO/codes/Models.py:126-139, 195-204:
 *     mmssl_mul_f32 / _bwd    bi_in = ego * side and its two gradients
 *     mmssl_ngcf_combine_f32  ego' = dropout(leaky_relu(G) + leaky_relu(B)) (keep: uint8 [rows, d] or NULL, kept entries
 *                             scaled), norm = ego' / max(|ego'|, eps) per row; _bwd: gradients of G and B from those of
 *                             ego' and norm (either may be NULL)
 *   d in {32, 64, 128, 256}; rows 16-byte aligned.
 * ---------------------------------------------------------------------------------- */
int mmssl_ell_spmm_f32(const int64_t* idx, const float* w, int64_t rows, int k, const float* H, int d, float* Y,
                       void* stream);
int mmssl_ell_spmm_bwd_f32(const int64_t* idx, const float* w, int64_t rows, int k, const float* H, int d,
                           const float* gY, float* gW, float* gH, void* stream);
int mmssl_mul_f32(const float* a, const float* b, int64_t n, float* out, void* stream);
int mmssl_mul_bwd_f32(const float* a, const float* b, const float* g, int64_t n, float* ga, float* gb, void* stream);
int mmssl_ngcf_combine_f32(const float* G, const float* B, const uint8_t* keep, float scale, int64_t rows, int d,
                           float eps, float* ego, float* norm, void* stream);
int mmssl_ngcf_combine_bwd_f32(const float* G, const float* B, const uint8_t* keep, float scale, const float* ego,
                               const float* g_ego, const float* g_norm, int64_t rows, int d, float eps, float* gG,
                               float* gB, void* stream);

/* ------------------------------------------------------------------------------------
 * Peer exchange (csrc/peer.hip): the row-sharded tables of mmssl_amd/dist.py move between the ranks of ONE node through
 * IPC-mapped device windows and epoch flags written by kernels - no collective library in the data path. The reference
 * has no multi-device path at all (MMSSL/main.py:529: one CUDA_VISIBLE_DEVICES); BASELINE.json's north_star asks for the
 * all-gather of neighbour embeddings before each propagation layer (MMSSL/Models.py:201-211), which this replaces RCCL for.
 *   context   one per process (= rank); `max_channels` epoch channels. Handles (mmssl_peer_handle_bytes() bytes each) travel
 *             between the processes by any host mechanism (the Python layer: torch.distributed all_gather_object).
 *   window    mmssl_peer_window_create allocates `bytes` of device memory (zeroed) and returns its IPC handle;
 *             mmssl_peer_window_open takes ALL ranks' handles of the same window id ([world][handle_bytes], own slot
 *             ignored) and maps the peers' buffers. Windows live until mmssl_peer_destroy.
 *   channel   mmssl_peer_signal: epoch[ch] += 1 and a system-scope store of it into slot [ch][rank] of every
 *             rank's flags; mmssl_peer_wait: returns (in stream order) once every slot [ch][*] of THIS rank's flags has
 *             reached epoch[ch] - i.e. every rank has signalled as often as this one. A wait that is not satisfied within
 *             the timeout (default 20 s) sets the context's error word and returns: the device is never hung;
 *             mmssl_peer_error copies the word to the host (blocking) - non-zero = some wait gave up, results invalid.
 *   push      rows [0, rows) x `width` floats of `src` (row pitch src_pitch floats) -> rows [dst_row0, ...) of EVERY rank's
 *             window `win_id` (row pitch dst_pitch), then signal(ch): the all-gather, pushed over every link at once.
 *             `wait_after` != 0: the launch's last block then also waits on the channel (push + wait in ONE launch).
 *             mmssl_peer_signal_wait: signal + wait as one launch (one wave).
 *   pull-sum  out[r] = sum_{q = 0 .. world-1, in that order} window_q[row0 + r]: the reduce-scatter as a pull with a
 *             fixed summation order (the same bits on every run and for every rank count's partition of the same sum
 *             order); call after signal + wait on the channel that guards the window.
 *   sum-slots out[j] = sum_{q = 0 .. n-1} slots[q * stride + j] (local): the second half of an all-reduce by push.
 *   Visibility (gfx942 / gfx950): pushed rows are stored write-through at system scope and counted with vmcnt before the
 *             epoch is published; data a plain kernel wrote into a window is published by that kernel's end (signal runs
 *             behind it in stream order); pull-sum / sum-slots read the windows past the caches; any other kernel that
 *             reads a window after mmssl_peer_wait must be a separate launch behind the wait (its start drops the caches'
 *             stale copies). No cache-wide fence inside the data kernels; the wait is one wave (a data kernel that spins
 *             in all of its blocks starves the kernels its peers wait for).
 * All compute calls are asynchronous on `stream` and hipGraph-capturable (epochs live in device memory).
 * widths, pitches: multiples of 4 floats; pointers 16-byte aligned; world <= 16.
 * ---------------------------------------------------------------------------------- */
typedef struct mmssl_peer mmssl_peer;
int mmssl_peer_create(int world, int rank, int max_channels, mmssl_peer** out);
int mmssl_peer_destroy(mmssl_peer* p);
/* info[0..5] = world, rank, max_channels, flags in fine-grained memory (0/1), windows, window bytes */
int mmssl_peer_info(const mmssl_peer* p, int64_t* info);
int mmssl_peer_set_timeout_ms(mmssl_peer* p, int64_t ms);
int mmssl_peer_handle_bytes(void);
int mmssl_peer_flags_handle(mmssl_peer* p, void* handle_out);
int mmssl_peer_open_flags(mmssl_peer* p, const void* handles);
int mmssl_peer_window_create(mmssl_peer* p, int64_t bytes, int* win_id, void* handle_out, void** local_ptr);
int mmssl_peer_window_open(mmssl_peer* p, int win_id, const void* handles);
int mmssl_peer_push_rows_f32(mmssl_peer* p, int ch, int win_id, const float* src, int64_t src_pitch, int64_t rows, int width,
                             int64_t dst_row0, int64_t dst_pitch, int wait_after, void* stream);
int mmssl_peer_signal(mmssl_peer* p, int ch, void* stream);
int mmssl_peer_signal_wait(mmssl_peer* p, int ch, void* stream);
int mmssl_peer_wait(mmssl_peer* p, int ch, void* stream);
int mmssl_peer_pull_sum_rows_f32(mmssl_peer* p, int win_id, int64_t row0, int64_t rows, int width, int64_t pitch, float* out,
                                 int64_t out_pitch, void* stream);
int mmssl_peer_sum_slots_f32(const float* slots, int n, int64_t stride, int64_t len, float* out, void* stream);
int mmssl_peer_error(mmssl_peer* p, uint32_t* err);

#ifdef __cplusplus
}
#endif
#endif /* MMSSL_HIP_H */
