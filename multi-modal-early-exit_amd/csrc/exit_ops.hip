// Exit heads' output projection, the on-device exit decision (confidence test -> wavefront ballot -> prefix sum ->
// stream compaction of the document list), row-map expansion, and the policy / threshold-sweep kernels.
//
// What this replaces in the reference:
//   * LayoutLMv3Exit.out_proj (EE/models/LayoutLMv3.py:92) / classifier.out_proj (HF:821) : head_out_kernel
//   * max_confidence / entropy criteria (EE/models/EE_modules.py:149-160)                   : crit_f32 / crit_f64
//   * Policy.max_confidence_global_thresholding_policy / accuracy_calibration_heuristic (EE/policy.py:28-45, 87-104):
//     the nested Python loop "first exit whose float64 max-softmax is strictly above its threshold, else the last"
//     becomes (a) exit_decide_kernel inside the forward pass — documents that satisfy the test are scattered to the
//     outputs and removed, deeper layers run on the survivors only — and (b) policy_scan_kernel on a dumped
//     (E+1,N,K) array, bit-identical in its integer outputs.
//   * thresh.opt0_2D / large_scale.check_2D_threshold (EE/thresh.py:184-215, EE/large_scale.py:42-84): threshold_sweep.
// All of it is HBM/latency-bound integer + small-vector work; none of it is shaped into a GEMM.
#include "mmee_kernels.h"

namespace mmee {

// ---------------------------------------------------------------------------------------------------------------
// out[i][c] = <in[row(i)], W[c]> + b[c]; one wave per document
// ---------------------------------------------------------------------------------------------------------------
template <int WAVES>
__device__ __forceinline__ void head_out_body(const HeadOutArgs& a) {
    const int n = *a.n_docs_ptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = blockIdx.x * WAVES + wave; i < n; i += gridDim.x * WAVES) {
        const int row = a.gather ? a.gather[i] : i;
        const float* x = a.in + (size_t)row * a.ld;
        f32x4 xv[kMaxNV];
#pragma unroll
        for (int k = 0; k < kMaxNV; ++k) {
            const int c = 4 * lane + 256 * k;
            xv[k] = (c < a.H) ? *reinterpret_cast<const f32x4*>(x + c) : f32x4{0, 0, 0, 0};
        }
        for (int o = 0; o < a.Ko; ++o) {
            const float* w = a.W + (size_t)o * a.H;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxNV; ++k) {
                const int c = 4 * lane + 256 * k;
                if (c < a.H) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
                    s += (xv[k][0] * wv[0] + xv[k][1] * wv[1]) + (xv[k][2] * wv[2] + xv[k][3] * wv[3]);
                }
            }
            s = wave_sum(s);
            if (lane == 0) a.out[(size_t)i * a.Ko + o] = s + a.b[o];
        }
    }
}

__global__ __launch_bounds__(256) void head_out_kernel(HeadOutArgs a) { head_out_body<4>(a); }

void launch_head_out(const HeadOutArgs& a, int max_docs, hipStream_t s) {
    int grid = (max_docs + 3) / 4;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(head_out_kernel, dim3(grid), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------------------
// criteria
// ---------------------------------------------------------------------------------------------------------------
// float32, as the model computes exit_states[j][1] (EE/models/EE_modules.py:149-160)
__device__ inline float crit_f32(const float* z, int K, int criterion) {
    if (criterion == 0) {
        float m = z[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, z[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(z[k] - m);
        return 1.0f / s;
    }
    float A = 0.f, B = 0.f;                       // entropy: log(sum e^x) - sum(x e^x)/sum(e^x), no max shift
    for (int k = 0; k < K; ++k) {
        const float e = expf(z[k]);
        A += e;
        B += z[k] * e;
    }
    return logf(A) - B / A;
}

// float64 on (double)logit / T, as the policy computes it (scipy.special.softmax on the float64 store, EE/policy.py:30-32)
__device__ inline double crit_f64(const float* z, int K, double temp, int criterion) {
    if (criterion == 0) {
        double m = (double)z[0] / temp;
        for (int k = 1; k < K; ++k) m = fmax(m, (double)z[k] / temp);
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += exp((double)z[k] / temp - m);
        return 1.0 / s;
    }
    double A = 0.0, B = 0.0;
    for (int k = 0; k < K; ++k) {
        const double x = (double)z[k] / temp;
        const double e = exp(x);
        A += e;
        B += x * e;
    }
    return log(A) - B / A;
}

// ---------------------------------------------------------------------------------------------------------------
// The exit stage: one workgroup of 1024 threads walks the active documents in chunks of 1024.
//   thread <-> document: criterion (f64), exit test, scatter of leavers to the output arrays;
//   survivors: wave ballot -> popcount prefix -> cross-wave prefix in LDS -> running carry = new dense index,
//   and the same scan over row counts = new dense row offset.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void exit_decide_body(const DecideArgs& a) {
    __shared__ int s_cnt[16], s_rows[16];
    __shared__ unsigned long long s_sq[16];
    __shared__ int s_carry_docs, s_carry_rows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.counts->n_docs;
    const double thr = a.thr_ptr ? a.thr_ptr[a.exit_index] : a.thr;
    const double temp = a.temp_ptr ? a.temp_ptr[a.exit_index] : a.temp;
    if (tid == 0) { s_carry_docs = 0; s_carry_rows = 0; }
    unsigned long long sq = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const bool active = i < n;
        bool keep = false;
        int len = 0, orig = 0;
        if (active) {
            orig = a.doc_orig[i];
            len = a.doc_off[i + 1] - a.doc_off[i];
            const float* z = a.pol_logits + (size_t)i * a.K;
            const double crit = crit_f64(z, a.K, temp, a.criterion);
            bool leave = a.criterion == 0 ? (crit > thr) : (crit < thr);   // strict, EE/policy.py:33
            if (a.no_exit) leave = false;
            if (a.is_final) leave = true;
            if (a.out_all_logits) {
                float* o = a.out_all_logits + ((size_t)a.exit_index * a.B + orig) * a.K;
                for (int k = 0; k < a.K; ++k) o[k] = (float)((double)z[k] / temp);
            }
            if (a.out_all_crit) a.out_all_crit[(size_t)a.exit_index * a.B + orig] = (float)crit;
            if (a.head_logits && a.out_head_logits) {
                const float* hz = a.head_logits + (size_t)i * a.Kh;
                float* o = a.out_head_logits + ((size_t)a.exit_index * a.B + orig) * a.Kh;
                for (int k = 0; k < a.Kh; ++k) o[k] = hz[k];
            }
            if (a.head_logits && a.out_head_crit)
                a.out_head_crit[(size_t)a.exit_index * a.B + orig] =
                    crit_f32(a.head_logits + (size_t)i * a.Kh, a.Kh, a.criterion);
            if (leave) {
                if (a.out_logits)
                    for (int k = 0; k < a.K; ++k) a.out_logits[(size_t)orig * a.K + k] = (float)((double)z[k] / temp);
                a.out_exit[orig] = a.exit_index;
                if (a.out_conf) a.out_conf[orig] = (float)crit;
            }
            keep = !leave;
        }
        // ---- wavefront ballot + prefix sums ------------------------------------------------------------------
        const unsigned long long ballot = __ballot(keep);
        const int before = __popcll(ballot & ((1ull << lane) - 1ull));       // survivors in lower lanes
        const int klen = keep ? len : 0;
        const int rows_incl = wave_incl_scan(klen, lane);
        if (lane == 63) { s_cnt[wave] = __popcll(ballot); s_rows[wave] = rows_incl; }
        if (keep) sq += (unsigned long long)len * (unsigned long long)len;
        __syncthreads();
        int wdocs = 0, wrows = 0, tdocs = 0, trows = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) { wdocs += s_cnt[w]; wrows += s_rows[w]; }
            tdocs += s_cnt[w];
            trows += s_rows[w];
        }
        const int cd = s_carry_docs, cr = s_carry_rows;
        if (keep) {
            const int k = cd + wdocs + before;
            a.n_doc_orig[k] = orig;
            a.n_doc_off[k] = cr + wrows + rows_incl - len;
            a.n_x_src[k] = a.x_phys[i];
            a.n_meta_src[k] = a.doc_off[i];
        }
        __syncthreads();
        if (tid == 0) { s_carry_docs = cd + tdocs; s_carry_rows = cr + trows; }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) s_sq[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_sq[w];
        a.n_doc_off[s_carry_docs] = s_carry_rows;
        a.n_counts->n_docs = s_carry_docs;
        a.n_counts->n_rows = s_carry_rows;
        a.n_counts->sum_len_sq = t;
    }
}

__global__ __launch_bounds__(1024) void exit_decide_kernel(DecideArgs a) { exit_decide_body(a); }

void launch_decide(const DecideArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(exit_decide_kernel, dim3(1), dim3(1024), 0, s, a);
}

// Round 6: the head's output projection and the decision were built as ONE launch (every workgroup writes its documents' logits, fences,
// takes a ticket; the last arrival runs the scan: nobody waits, same bits, 91 GPU tests green) and measured on config 3 (2 x 512 documents, 48 exits per step):
// 52.7 us per exit against 31 + 16 us for the two launches -- the fences and the serial scan behind the last arrival cost what the launch saved.  Removed.

// new dense row r of surviving document k  <-  physical X row n_x_src[k] + t, metadata row n_meta_src[k] + t
__global__ __launch_bounds__(256) void compact_rows_kernel(const StageCounts* n_counts, const int* __restrict__ n_doc_off,
                                                           const int* __restrict__ n_x_src, const int* __restrict__ n_meta_src,
                                                           const RowMeta* __restrict__ meta_old, RowMeta* __restrict__ meta_new,
                                                           int* __restrict__ row_src) {
    const int n = n_counts->n_docs;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int off = n_doc_off[k], len = n_doc_off[k + 1] - off;
        const int xs = n_x_src[k], ms = n_meta_src[k];
        for (int t = threadIdx.x; t < len; t += 256) {
            row_src[off + t] = xs + t;
            meta_new[off + t] = meta_old[ms + t];
        }
    }
}

void launch_compact_rows(const StageCounts* n_counts, const int* n_doc_off, const int* n_x_src, const int* n_meta_src,
                         const RowMeta* meta_old, RowMeta* meta_new, int* row_src, int max_docs, int num_cus, hipStream_t s) {
    int grid = max_docs < num_cus * 8 ? max_docs : num_cus * 8;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(compact_rows_kernel, dim3(grid), dim3(256), 0, s, n_counts, n_doc_off, n_x_src, n_meta_src,
                       meta_old, meta_new, row_src);
}

// (logits f32 (n,K), exit_layer i32 (n), confidence f32 (n)) <-> the row of the ONE all-gather of the north star: K + 2 int32 words per document
// (the floats travel as their bit patterns: integer copies and collectives never flush, canonicalise or round them)
__global__ __launch_bounds__(256) void pack_results_kernel(const float* __restrict__ logits, const int* __restrict__ exit_layer,
                                                           const float* __restrict__ conf, int n, int K, int* __restrict__ rows) {
    const long total = (long)n * (K + 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d = (int)(i / (K + 2)), c = (int)(i - (long)d * (K + 2));
        rows[i] = c < K ? __float_as_int(logits[(size_t)d * K + c]) : c == K ? exit_layer[d] : __float_as_int(conf[d]);
    }
}
__global__ __launch_bounds__(256) void unpack_results_kernel(const int* __restrict__ rows, int n, int K, float* __restrict__ logits,
                                                             int* __restrict__ exit_layer, float* __restrict__ conf) {
    const long total = (long)n * (K + 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d = (int)(i / (K + 2)), c = (int)(i - (long)d * (K + 2));
        const int v = rows[i];
        if (c < K) { if (logits) logits[(size_t)d * K + c] = __int_as_float(v); }
        else if (c == K) { if (exit_layer) exit_layer[d] = v; }
        else if (conf) conf[d] = __int_as_float(v);
    }
}
void launch_pack_results(const float* logits, const int* exit_layer, const float* conf, int n, int K, int* rows, hipStream_t s) {
    long total = (long)n * (K + 2);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(pack_results_kernel, dim3(grid), dim3(256), 0, s, logits, exit_layer, conf, n, K, rows);
}
void launch_unpack_results(const int* rows, int n, int K, float* logits, int* exit_layer, float* conf, hipStream_t s) {
    long total = (long)n * (K + 2);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(unpack_results_kernel, dim3(grid), dim3(256), 0, s, rows, n, K, logits, exit_layer, conf);
}

// out[orig][:] = X[x_phys[i]][:]   (CLS rows, parity/debug output)
// CLS row of every active document -> out[doc_orig ? doc_orig[i] : i].  split_inv != 0: X holds split-f16 rows (1 / scale = split_inv)
__global__ __launch_bounds__(256) void gather_cls_kernel(const float* __restrict__ X, int H, const int* __restrict__ x_phys,
                                                         const int* __restrict__ doc_orig, const int* __restrict__ n_docs_ptr,
                                                         float* __restrict__ out, float split_inv) {
    const int n = *n_docs_ptr;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float* src = X + (size_t)x_phys[i] * H;
        float* dst = out + (size_t)(doc_orig ? doc_orig[i] : i) * H;
        if (split_inv != 0.f) {
            for (int c = 4 * threadIdx.x; c < H; c += 1024) *reinterpret_cast<f32x4*>(dst + c) = load_split4(src, c, split_inv);
        } else {
            for (int c = threadIdx.x; c < H; c += 256) dst[c] = src[c];
        }
    }
}

// Hidden states in the reference's padded layout (ee_set_hidden_states_out): position p of document d is text token p (packed row
// text_dst[d * T + p], < 0 when the ragged layout dropped it: zeros) or visual row p - T (packed behind the document's ntext[d] text rows).
__global__ __launch_bounds__(256) void rows_to_padded_kernel(const float* __restrict__ X, float split_inv, int H, int B, int T, int Pv,
                                                             const int* __restrict__ text_dst, const int* __restrict__ ntext,
                                                             const int* __restrict__ doc_off, float* __restrict__ out) {
    const int S = T + Pv;
    for (long r = blockIdx.x; r < (long)B * S; r += gridDim.x) {
        const int d = (int)(r / S), p = (int)(r - (long)d * S);
        int row;
        if (p < T) {
            const int t = text_dst[(size_t)d * T + p];
            row = t < 0 ? -1 : doc_off[d] + t;
        } else {
            row = doc_off[d] + (text_dst ? ntext[d] : 0) + (p - T);
        }
        float* dst = out + (size_t)r * H;
        if (row < 0) {
            for (int c = threadIdx.x; c < H; c += 256) dst[c] = 0.f;
        } else if (split_inv != 0.f) {
            const float* src = X + (size_t)row * H;
            for (int c = 4 * threadIdx.x; c < H; c += 1024) *reinterpret_cast<f32x4*>(dst + c) = load_split4(src, c, split_inv);
        } else {
            const float* src = X + (size_t)row * H;
            for (int c = threadIdx.x; c < H; c += 256) dst[c] = src[c];
        }
    }
}
void launch_rows_to_padded(const float* X, float split_inv, int H, int B, int T, int Pv, const int* text_dst, const int* ntext, const int* doc_off,
                           float* out, hipStream_t s) {
    long rows = (long)B * (T + Pv);
    int grid = rows < 4096 ? (int)rows : 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(rows_to_padded_kernel, dim3(grid), dim3(256), 0, s, X, split_inv, H, B, T, Pv, text_dst, ntext, doc_off, out);
}

void launch_gather_cls(const float* X, int H, const int* x_phys, const int* doc_orig, const int* n_docs_ptr, float* out,
                       int max_docs, hipStream_t s, float split_inv) {
    int grid = max_docs < 2048 ? max_docs : 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gather_cls_kernel, dim3(grid), dim3(256), 0, s, X, H, x_phys, doc_orig, n_docs_ptr, out, split_inv);
}

// ---------------------------------------------------------------------------------------------------------------
// Policy on a dumped (E1, N, K) float64 array: thread per document, scan exits in order.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void policy_scan_kernel(const double* __restrict__ logits, int E1, int N, int K,
                                                          const double* __restrict__ thr, int* __restrict__ exits,
                                                          double* __restrict__ pred, double* __restrict__ conf_out,
                                                          int* __restrict__ counts) {
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
        int chosen = E1 - 1;
        double cchosen = 0.0;
        for (int e = 0; e < E1; ++e) {
            const double* z = logits + ((size_t)e * N + n) * K;
            double m = z[0];
            for (int k = 1; k < K; ++k) m = fmax(m, z[k]);
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += exp(z[k] - m);
            const double c = 1.0 / s;
            cchosen = c;
            if (c > thr[e]) { chosen = e; break; }
        }
        exits[n] = chosen;
        if (conf_out) conf_out[n] = cchosen;
        if (pred) {
            const double* z = logits + ((size_t)chosen * N + n) * K;
            for (int k = 0; k < K; ++k) pred[(size_t)n * K + k] = z[k];
        }
        if (counts) atomicAdd(&counts[chosen], 1);
    }
}

void launch_policy_scan(const double* logits, int E1, int N, int K, const double* thr_dev, int* exits, double* pred,
                        double* conf, int* counts, hipStream_t s) {
    int grid = (N + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(policy_scan_kernel, dim3(grid), dim3(256), 0, s, logits, E1, N, K, thr_dev, exits, pred, conf, counts);
}

// ---------------------------------------------------------------------------------------------------------------
// Threshold sweep: one workgroup per threshold vector, documents strided over its 256 threads.
//   exit(v, n) = first e with conf[e][n] >= thr[v][e], else 0 (numpy argmax of an all-False column)
// conf is (E1, N) so consecutive threads read consecutive documents of the same exit row (coalesced).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void threshold_sweep_kernel(const double* __restrict__ conf, const unsigned char* __restrict__ correct,
                                                              int E1, int N, const double* __restrict__ thr, int V,
                                                              double* __restrict__ acc, double* __restrict__ mean_exit,
                                                              int* __restrict__ hist) {
    __shared__ double s_thr[64];
    __shared__ int s_hist[64];
    __shared__ unsigned long long s_red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int v = blockIdx.x; v < V; v += gridDim.x) {
        __syncthreads();
        if (tid < E1) { s_thr[tid] = thr[(size_t)v * E1 + tid]; s_hist[tid] = 0; }
        __syncthreads();
        unsigned int n_correct = 0, sum_exit = 0;
        for (int n = tid; n < N; n += 256) {
            int ex = 0;
            for (int e = 0; e < E1; ++e) {
                if (conf[(size_t)e * N + n] >= s_thr[e]) { ex = e; break; }
            }
            n_correct += correct[(size_t)ex * N + n];
            sum_exit += ex;
            if (hist) atomicAdd(&s_hist[ex], 1);
        }
        unsigned long long packed = ((unsigned long long)n_correct << 32) | (unsigned long long)sum_exit;
        for (int o = 32; o > 0; o >>= 1) packed += __shfl_xor(packed, o, 64);
        if (lane == 0) s_red[wave] = packed;
        __syncthreads();
        if (tid == 0) {
            const unsigned long long t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            acc[v] = (double)(t >> 32) / (double)N;
            mean_exit[v] = (double)(t & 0xffffffffull) / (double)N;
        }
        if (hist && tid < E1) hist[(size_t)v * E1 + tid] = s_hist[tid];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The sweep at the reference's scale (EE/large_scale.py:46-84 with num_mixtures = 1 500 000 threshold vectors, :179-180).  The kernel above
// re-reads the whole (E1, N) float64 table for every vector (2.2 MB x 1.5 M = 3.4 TB of cache traffic) and compares doubles.  Only the
// ORDER of conf[e][n] and thr[v][e] matters, so the comparison is done on integer ranks instead:
//     p[e][n] = #{m : conf[e][m] < conf[e][n]},   t[v][e] = #{m : conf[e][m] < thr[v][e]}      =>      conf[e][n] >= thr[v][e]  <=>  p[e][n] >= t[v][e]
// (>=: every element below thr is below conf, so t <= p; <: conf itself and everything below it is below thr, so t >= p + 1).  Exact for any
// doubles, ties and duplicates included (NaN confidences do not occur: they are softmax maxima).
//   1. sweep_rank_kernel   p by counting (N^2 / exit, 1.1e10 double compares at 7 x 40 000: ~1 ms) and, with the tie index from the same pass,
//                          the sorted confidences of every exit;
//   2. sweep_thr_kernel    t by binary search in the sorted row;
//   3. sweep_main_kernel   one THREAD per threshold vector (its E1 ranks in registers, its two sums in registers: no reduction across
//                          lanes), the documents streamed through LDS as records  rec[e] = p << 8 | correct << 6 | e  that every lane reads
//                          at the same address (broadcast).  exit = first e with rec[e] >= t[e] << 8, else 0 (numpy argmax of an all-False
//                          column): r = rec[0]; for e = E1 - 1 .. 0: r = rec[e] >= T[e] ? rec[e] : r  -- two vector instructions per exit --
//                          and the payload bits of r give (correct, exit).  Integer work per (vector, document): 2 E1 + 4 instructions.
// The table is read once per 256 vectors from L2 (1.3 MB x V / 256).  Needs N < 2^24 and E1 <= 64; the histogram output stays on the kernel above.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_rank_kernel(const double* __restrict__ conf, const unsigned char* __restrict__ correct, int E1, int E1P,
                                                         int N, unsigned* __restrict__ rec, double* __restrict__ sorted) {
    __shared__ double tile[2048];
    const int e = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    const double* row = conf + (size_t)e * N;
    const double c = n < N ? row[n] : 0.0;
    unsigned lt = 0, eq_before = 0;
    for (int m0 = 0; m0 < N; m0 += 2048) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2048; i += 256) tile[i] = m0 + i < N ? row[m0 + i] : 0.0;
        __syncthreads();
        const int cnt = N - m0 < 2048 ? N - m0 : 2048;
        for (int i = 0; i < cnt; ++i) {
            const double x = tile[i];
            lt += x < c ? 1u : 0u;
            eq_before += (x == c && m0 + i < n) ? 1u : 0u;
        }
    }
    if (n < N) {
        rec[(size_t)n * E1P + e] = (lt << 8) | ((unsigned)(correct[(size_t)e * N + n] ? 1u : 0u) << 6) | (unsigned)e;
        sorted[(size_t)e * N + lt + eq_before] = c;                  // equal values take consecutive places in document order
    }
}

__global__ __launch_bounds__(256) void sweep_thr_kernel(const double* __restrict__ sorted, const double* __restrict__ thr, int E1, int N, long long VE,
                                                        unsigned* __restrict__ T) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < VE; i += (long long)gridDim.x * 256) {
        const int e = (int)(i % E1);
        const double t = thr[i];
        const double* row = sorted + (size_t)e * N;
        int lo = 0, hi = N;                                          // first index with row[idx] >= t  =  number of confidences below t
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (row[mid] < t) lo = mid + 1;
            else hi = mid;
        }
        T[i] = t != t ? 0xffffffffu : (unsigned)lo << 8;             // conf >= NaN is false for every document (numpy): a word no record reaches
    }
}

template <int E1C>      // E1C > 0: compile-time exit count (unrolled, ranks in registers); 0: run-time E1 (ranks re-read from the vector's row)
__global__ __launch_bounds__(256, 2) void sweep_main_kernel(const unsigned* __restrict__ rec, const unsigned* __restrict__ T, int E1, int E1P, int N,
                                                            int V, double* __restrict__ acc, double* __restrict__ mean_exit) {
    extern __shared__ unsigned s_rec[];                              // CHUNK documents x E1P words
    const int chunk = (64 * 1024) / (4 * E1P);
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int vv = v < V ? v : V - 1;
    unsigned tq[E1C > 0 ? E1C : 1];
    if (E1C > 0) {
#pragma unroll
        for (int e = 0; e < E1C; ++e) tq[e] = T[(size_t)vv * E1 + e];
    }
    unsigned n_correct = 0, sum_exit = 0;
    for (int n0 = 0; n0 < N; n0 += chunk) {
        const int cnt = N - n0 < chunk ? N - n0 : chunk;
        __syncthreads();
        {
            const uint4* src = reinterpret_cast<const uint4*>(rec + (size_t)n0 * E1P);
            uint4* dst = reinterpret_cast<uint4*>(s_rec);
            const int n16 = cnt * E1P / 4;
            for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
        }
        __syncthreads();
        if (E1C > 0) {
#pragma unroll 4
            for (int i = 0; i < cnt; ++i) {
                const unsigned* d = s_rec + i * E1P;                 // the same address in every lane: a broadcast read
                const unsigned d0 = d[0];
                unsigned r = d0;                                     // no exit fires: exit 0
#pragma unroll
                for (int e = E1C - 1; e >= 1; --e) {
                    const unsigned x = d[e];
                    r = x >= tq[e] ? x : r;
                }
                r = d0 >= tq[0] ? d0 : r;                            // exit 0 fires: it is the first
                n_correct += (r >> 6) & 1u;
                sum_exit += r & 63u;
            }
        } else {
            for (int i = 0; i < cnt; ++i) {
                const unsigned* d = s_rec + i * E1P;
                const unsigned d0 = d[0];
                unsigned r = d0;
                for (int e = E1 - 1; e >= 1; --e) {
                    const unsigned x = d[e];
                    r = x >= T[(size_t)vv * E1 + e] ? x : r;
                }
                r = d0 >= T[(size_t)vv * E1] ? d0 : r;
                n_correct += (r >> 6) & 1u;
                sum_exit += r & 63u;
            }
        }
    }
    if (v < V) {
        acc[v] = (double)n_correct / (double)N;
        mean_exit[v] = (double)sum_exit / (double)N;
    }
}

void launch_threshold_sweep(const double* conf, const unsigned char* correct, int E1, int N, const double* thr, int V,
                            double* acc, double* mean_exit, int* hist, hipStream_t s) {
    // ranks: the integer sweep (no histogram, N < 2^24, enough vectors to pay for the O(N^2) ranking pass)
    const bool ranked = !hist && N < (1 << 24) && E1 <= 64 && (long long)V * 8 >= (long long)N;
    if (ranked) {
        const int E1P = (E1 + 3) & ~3;
        unsigned *rec = nullptr, *T = nullptr;
        double* sorted = nullptr;
        if (hipMallocAsync((void**)&rec, (size_t)N * E1P * 4, s) == hipSuccess && hipMallocAsync((void**)&T, (size_t)V * E1 * 4, s) == hipSuccess &&
            hipMallocAsync((void**)&sorted, (size_t)E1 * N * 8, s) == hipSuccess) {
            (void)hipMemsetAsync(rec, 0, (size_t)N * E1P * 4, s);
            hipLaunchKernelGGL(sweep_rank_kernel, dim3((N + 255) / 256, E1), dim3(256), 0, s, conf, correct, E1, E1P, N, rec, sorted);
            const long long VE = (long long)V * E1;
            int g2 = (int)((VE + 255) / 256 < 65536 ? (VE + 255) / 256 : 65536);
            hipLaunchKernelGGL(sweep_thr_kernel, dim3(g2), dim3(256), 0, s, sorted, thr, E1, N, VE, T);
            const int grid = (V + 255) / 256;
            const size_t lds = 64 * 1024;
            (void)ensure_dynamic_lds<&sweep_main_kernel<7>>("sweep_main_kernel", (int)lds);
            (void)ensure_dynamic_lds<&sweep_main_kernel<0>>("sweep_main_kernel", (int)lds);
            if (E1 == 7) hipLaunchKernelGGL((sweep_main_kernel<7>), dim3(grid), dim3(256), lds, s, rec, T, E1, E1P, N, V, acc, mean_exit);
            else hipLaunchKernelGGL((sweep_main_kernel<0>), dim3(grid), dim3(256), lds, s, rec, T, E1, E1P, N, V, acc, mean_exit);
            (void)hipFreeAsync(rec, s); (void)hipFreeAsync(T, s); (void)hipFreeAsync(sorted, s);
            return;
        }
        (void)hipGetLastError();                                     // allocation failed: the direct kernel needs no workspace
        if (rec) (void)hipFreeAsync(rec, s);
        if (T) (void)hipFreeAsync(T, s);
        if (sorted) (void)hipFreeAsync(sorted, s);
    }
    int grid = V < 8192 ? V : 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(threshold_sweep_kernel, dim3(grid), dim3(256), 0, s, conf, correct, E1, N, thr, V, acc, mean_exit, hist);
}

// conf[e][n] = max softmax (f64) of logits[e][n][:], correct[e][n] = (argmax == reference[n])   (first maximum wins, as numpy)
__global__ __launch_bounds__(256) void msp_table_kernel(const double* __restrict__ logits, const long long* __restrict__ refs,
                                                        int E1, int N, int K, double* __restrict__ conf,
                                                        unsigned char* __restrict__ correct) {
    const size_t total = (size_t)E1 * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const double* z = logits + i * K;
        double m = z[0];
        int am = 0;
        for (int k = 1; k < K; ++k)
            if (z[k] > m) { m = z[k]; am = k; }
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += exp(z[k] - m);
        conf[i] = 1.0 / s;
        if (correct) correct[i] = refs ? (unsigned char)(refs[i % N] == am) : 0;
    }
}

void launch_msp_table(const double* logits, const long long* refs, int E1, int N, int K, double* conf, unsigned char* correct,
                      hipStream_t s) {
    size_t total = (size_t)E1 * N;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(msp_table_kernel, dim3(grid), dim3(256), 0, s, logits, refs, E1, N, K, conf, correct);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-exit temperature fit: argmin_T mean NLL(softmax(z / T), y)   (TemperatureScaler.set_temperature,
// EE/generic_scaling.py:64-111, L-BFGS-B on sklearn log_loss).  In beta = 1/T the objective
//     f(beta) = mean( logsumexp(beta z) - beta z_y )
// is convex with f' = mean(E_p[z] - z_y) and f'' = mean(Var_p[z]) (p = softmax(beta z)), so a safeguarded Newton
// iteration converges in a handful of steps; one workgroup per exit keeps the whole loop on the device.
// Also returns the quantities calibrate() derives from the scaled logits (EE/eval.py:313-337): NLL, accuracy
// (argmax == label) and mean max-softmax confidence at the fitted temperature.
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void block_sum3(double& a, double& b, double& c, double* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
        c += __shfl_xor(c, o, 64);
    }
    __syncthreads();
    if (lane == 0) { lds[3 * wave] = a; lds[3 * wave + 1] = b; lds[3 * wave + 2] = c; }
    __syncthreads();
    a = b = c = 0.0;
    for (int w = 0; w < nw; ++w) { a += lds[3 * w]; b += lds[3 * w + 1]; c += lds[3 * w + 2]; }
}

__global__ __launch_bounds__(1024) void temperature_fit_kernel(const double* __restrict__ logits, const long long* __restrict__ labels,
                                                               int N, int K, int max_iter, double* __restrict__ T_out,
                                                               double* __restrict__ nll_out, double* __restrict__ acc_out,
                                                               double* __restrict__ conf_out, int* __restrict__ iters_out) {
    __shared__ double red[3 * 16];
    const int e = blockIdx.x;
    const double* L = logits + (size_t)e * N * K;
    double beta = 1.0;                                    // TemperatureScaler starts from T = 1 (generic_scaling.py:42-46)
    int it = 0;
    for (; it < max_iter; ++it) {
        double g = 0.0, h = 0.0, f = 0.0;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            const double* z = L + (size_t)n * K;
            double m = z[0];
            for (int k = 1; k < K; ++k) m = fmax(m, z[k]);
            double S = 0.0, A = 0.0, B = 0.0;
            for (int k = 0; k < K; ++k) {
                const double w = exp(beta * (z[k] - m));
                S += w; A += w * z[k]; B += w * z[k] * z[k];
            }
            A /= S; B /= S;
            const double zy = z[labels[n]];
            g += A - zy;
            h += B - A * A;
            f += log(S) + beta * (m - zy);
        }
        block_sum3(g, h, f, red);
        g /= N; h /= N;
        double step = (h > 1e-300) ? g / h : (g > 0 ? 0.5 * beta : -beta);
        double nb = beta - step;
        if (!(nb > 0.0)) nb = 0.5 * beta;                 // stay in beta > 0 (the reference bounds T in (1e-32, inf))
        if (nb > 64.0 * beta) nb = 64.0 * beta;
        const bool done = fabs(nb - beta) <= 1e-13 * beta;
        beta = nb;
        if (done) break;
    }
    // final statistics at the fitted temperature
    double f = 0.0, acc = 0.0, conf = 0.0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const double* z = L + (size_t)n * K;
        double m = z[0];
        int am = 0;
        for (int k = 1; k < K; ++k)
            if (z[k] > m) { m = z[k]; am = k; }
        double S = 0.0;
        for (int k = 0; k < K; ++k) S += exp(beta * (z[k] - m));
        f += log(S) + beta * (m - z[labels[n]]);
        acc += (am == (int)labels[n]) ? 1.0 : 0.0;
        conf += 1.0 / S;
    }
    block_sum3(f, acc, conf, red);
    if (threadIdx.x == 0) {
        T_out[e] = 1.0 / beta;
        if (nll_out) nll_out[e] = f / N;
        if (acc_out) acc_out[e] = acc / N;
        if (conf_out) conf_out[e] = conf / N;
        if (iters_out) iters_out[e] = it;
    }
}

void launch_temperature_fit(const double* logits, const long long* labels, int E1, int N, int K, int max_iter, double* T_out,
                            double* nll_out, double* acc_out, double* conf_out, int* iters_out, hipStream_t s) {
    hipLaunchKernelGGL(temperature_fit_kernel, dim3(E1), dim3(1024), 0, s, logits, labels, N, K, max_iter, T_out, nll_out,
                       acc_out, conf_out, iters_out);
}

// ---------------------------------------------------------------------------------------------------------------
// value tables of the relative-position bias: t[h][delta + c] = W[h][lut[delta + c]] / sqrt(d)
// ---------------------------------------------------------------------------------------------------------------
__global__ void build_value_tables_kernel(const float* w1, const float* wx, const float* wy, const unsigned char* lut1,
                                          const unsigned char* lut2, int heads, int bins1, int bins2, int n1, int n2,
                                          float inv_sqrt_d, float* t1, float* tx, float* ty) {
    const int h = blockIdx.x;
    for (int i = threadIdx.x; i < n1; i += blockDim.x) t1[(size_t)h * n1 + i] = w1[(size_t)h * bins1 + lut1[i]] * inv_sqrt_d;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        tx[(size_t)h * n2 + i] = wx[(size_t)h * bins2 + lut2[i]] * inv_sqrt_d;
        ty[(size_t)h * n2 + i] = wy[(size_t)h * bins2 + lut2[i]] * inv_sqrt_d;
    }
}

void launch_build_value_tables(const float* w1, const float* wx, const float* wy, const unsigned char* lut1,
                               const unsigned char* lut2, int heads, int bins1, int bins2, int n1, int n2, float inv_sqrt_d,
                               float* t1, float* tx, float* ty, hipStream_t s) {
    hipLaunchKernelGGL(build_value_tables_kernel, dim3(heads), dim3(256), 0, s, w1, wx, wy, lut1, lut2, heads, bins1, bins2,
                       n1, n2, inv_sqrt_d, t1, tx, ty);
}

}  // namespace mmee
