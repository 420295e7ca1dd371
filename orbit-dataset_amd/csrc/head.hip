// Prototype head (ProtoNets) + frame pooler for gfx950.
//
// Replaces, for the episodic hot path, the ATen op sequence under
//   model/classifier_heads.py:94-119   HeadClassifier._build_class_reps (unique -> index_select -> mean)
//   model/classifier_heads.py:232-263  PrototypicalClassifier.configure (W = 2 mu, b = -mu.mu)
//   model/classifier_heads.py:202-230  PrototypicalClassifier.predict (euclidean F.linear / cosine)
//   model/poolers.py:7-16              MeanPooler.forward (fused: T frames per clip are averaged on load)
//
// All three kernels are HBM-bound (AI ~ 2.4 FLOP/B): rows are streamed once with 16-byte loads, the
// class matrix W (C*D*4 B ~ 25 KB) stays in L1/L2, reductions are wave64 DPP/shuffle reductions.
#include <cstdlib>
#include "common.h"

namespace orbit {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- library options (tuning switches): name -> int; initial value from the environment ORBIT_<NAME upper-cased>
struct Option {
    const char* name;
    const char* env;
    int value;
    bool init;
};
static Option g_options[] = {
    // network runtime
    {"graph", "ORBIT_GRAPH", 2, false},              // forward launch sequences as HIP graphs: 0 never, 1 always, 2 adaptive
    {"train_graph", "ORBIT_TRAIN_GRAPH", 1, false},  // the same for the training entry points: 0 never, 1 from the third sight of a call
    {"mbconv_rows", "ORBIT_MBCONV_ROWS", 1, false},  // row-streaming fused MBConv fronts at plan creation (0 = conv + depthwise pair)
    {"stem_rows", "ORBIT_STEM_ROWS", 1, false},      // the same for stem + first depthwise
    {"train_dw_xf", "ORBIT_TRAIN_DW_XF", 1, false},  // no-backward training forwards: BatchNorm + SiLU applied on the depthwise load
    {"train_fused_fronts", "ORBIT_TRAIN_FUSED_FRONTS", 1, false},  // ... and MBConv fronts as statistics sweep + row-streaming kernel
    // dense convolutions
    {"conv_tile", "ORBIT_CONV_TILE", 0, false},      // 0 heuristic; 3 = 64x64, 4 = 128x32, 6 = 32x32 with K split over the waves
    {"conv_bk", "ORBIT_CONV_BK", 0, false},          // 0 = widest K-tile that divides Cin; 8 / 16 / 32 caps it
    {"conv_splitk", "ORBIT_CONV_SPLITK", 1, false},  // split-K over blocks for short, long-K layers
    {"conv_rgemm", "ORBIT_CONV_RGEMM", 1, false},    // pointwise register GEMM: 0 never, 1 where measured faster, 2 wherever supported
    {"conv_bf3", "ORBIT_CONV_BF3", 0, false},        // OPT-IN bf16 x 3 split (bit 1 dense convs, bit 2 fused-front expands); never in `value`
    // depthwise kernel families: 1 = where measured faster (default), 0 = never, 2 = wherever it fits
    {"dw_window", "ORBIT_DW_WINDOW", 1, false},
    {"dw_lds", "ORBIT_DW_LDS", 1, false},
    {"dw_pipe", "ORBIT_DW_PIPE", 1, false},
    // head
    {"head_stream", "ORBIT_HEAD_STREAM", 1, false}};  // streaming distance kernel (T = 1, D = 512 / 1280); 0 = general LDS form
static Option* find_option(const char* name) {
    for (Option& o : g_options)
        if (strcmp(o.name, name) == 0) {
            if (!o.init) {
                const char* e = getenv(o.env);
                if (e) o.value = atoi(e);
                o.init = true;
            }
            return &o;
        }
    return nullptr;
}
int get_option(const char* name) {
    Option* o = find_option(name);
    return o ? o->value : 0;
}
// bumped by every orbit_set_option that changes a value: captured launch sequences (csrc/extractor.hip, extractor.h) carry the
// epoch they were recorded under in their key, so a graph never replays kernels chosen under other option values
static int g_option_epoch = 0;
int option_epoch() { return g_option_epoch; }

// wave64 sum on the VALU with DPP lane permutes (quad swaps, half-row / row mirrors, then row broadcasts), result
// broadcast from lane 63. __shfl_xor lowers to ds_bpermute_b32, an LDS-pipe round trip per step: with 20 reductions per
// wave in the distance kernel those 120 dependent round trips, not HBM, set the kernel time.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the wave sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- configure: segmented sum by class --------------------------------------------------------
// grid (ceil(D/256), C, n_tasks), block 256: thread = one feature column d of one class.
// Walks the N clips in ascending order, so every (c, d) sum has a fixed order (deterministic, and the
// same order a sequential per-class mean would use). Coalesced: a wave reads 256 B of one row.
__global__ __launch_bounds__(256) void proto_configure_kernel(
    const float* __restrict__ feats, const int64_t* __restrict__ labels,
    const int64_t* __restrict__ class_ids, int N, int T, int D, int C,
    float* __restrict__ sums, float* __restrict__ counts) {
    const int task = blockIdx.z, c = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int64_t cid = class_ids[(size_t)task * C + c];
    const float* f = feats + (size_t)task * N * T * D;
    const int64_t* lab = labels + (size_t)task * N;
    const float invT = 1.0f / (float)T;
    // Round 6: the class's clips are first COMPACTED into an index list in LDS (ascending, chunks of 1024 labels: one coalesced
    // label load per thread and a block scan), then summed from the list. The first form tested `lab[i] != cid` inside the
    // row loop: N dependent label loads, each a wave-uniform L2 round trip, made a 200-clip configure a 25 us latency chain on
    // the critical path between the support pass and the head (per-launch events, bench.py roofline.families); with the list
    // the row loads of a class are independent of each other. Same rows, same ascending order: bit-identical sums.
    __shared__ int idx[1024];
    __shared__ int wave_cnt[4];
    __shared__ int chunk_n;
    float acc = 0.f;
    int cnt = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < N; base += 1024) {
        int mine[4], nm = 0;  // thread t owns labels base + 4 t .. + 3 (ascending inside the thread, threads ascending)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + 4 * (int)threadIdx.x + j;
            if (i < N && lab[i] == cid) mine[nm++] = i;
        }
        // exclusive scan of nm over the block: prefix inside the wave by shuffles, then the 4 wave totals
        int pre = nm;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(pre, off);
            if (lane >= off) pre += o;
        }
        if (lane == 63) wave_cnt[wave] = pre;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
        const int start = woff + pre - nm;
        for (int j = 0; j < nm; ++j) idx[start + j] = mine[j];
        if (threadIdx.x == 255) chunk_n = woff + pre;
        __syncthreads();
        const int n = chunk_n;
        cnt += n;
        if (d < D) {
            for (int k = 0; k < n; ++k) {
                const float* row = f + (size_t)idx[k] * T * D + d;
                if (T == 1) {
                    acc += row[0];
                } else {
                    float s = 0.f;
                    for (int t = 0; t < T; ++t) s += row[(size_t)t * D];
                    acc += s * invT;
                }
            }
        }
        __syncthreads();  // idx / wave_cnt are rewritten by the next chunk
    }
    if (d < D) sums[((size_t)task * C + c) * D + d] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(size_t)task * C + c] = (float)cnt;
}

// ---- label set: the ascending unique values of labels[N] (what torch.unique gives the reference's configure, reference
// model/classifier_heads.py:96-100,246-248) WITHOUT a host round trip: class_ids[0 .. cap) receives them (slots beyond the
// count repeat the last value, so every slot is a real class and kernels launched over `cap` slots compute finite rows),
// *count the number found, or cap + 1 when there are more than `cap`. One wave: selection by repeated minimum - the label
// sets of a task are 5..20 values over a few hundred clips.
__global__ __launch_bounds__(64) void label_set_kernel(const int64_t* __restrict__ labels, int N,
                                                       int64_t* __restrict__ class_ids, int cap, int32_t* __restrict__ count) {
    const int lane = threadIdx.x;
    long long last = 0;
    bool have = false;
    int found = 0;
    for (;;) {
        long long best = 0;
        int any = 0;
        for (int i = lane; i < N; i += 64) {
            const long long v = labels[i];
            if ((!have || v > last) && (!any || v < best)) best = v, any = 1;
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const long long ob = __shfl_xor(best, off);
            const int oa = __shfl_xor(any, off);
            if (oa && (!any || ob < best)) best = ob, any = 1;
        }
        if (!any) break;
        if (found < cap && lane == 0) class_ids[found] = best;
        ++found, last = best, have = true;
        if (found > cap) break;
    }
    if (lane == 0) {
        *count = found;
        for (int i = found; i < cap; ++i) class_ids[i] = have ? last : 0;
    }
}

// ---- finalize: W = 2 mu, b = -mu.mu ------------------------------------------------------------
// grid (C, n_tasks), block 256
__global__ __launch_bounds__(256) void proto_finalize_kernel(const float* __restrict__ sums,
                                                             const float* __restrict__ counts, int C,
                                                             int D, int cosine, float* __restrict__ W,
                                                             float* __restrict__ b) {
    __shared__ float red[4];
    const int task = blockIdx.y, c = blockIdx.x;
    const size_t base = ((size_t)task * C + c) * D;
    const float cnt = counts[(size_t)task * C + c];
    float sq = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float mu = sums[base + d] / cnt;
        W[base + d] = 2.0f * mu;
        sq += mu * mu;
    }
    if (cosine) return;
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) b[(size_t)task * C + c] = -(red[0] + red[1] + red[2] + red[3]);
}

// ---- predict -----------------------------------------------------------------------------------
// One wave handles R consecutive query clips; lanes stride the feature dimension with float4 loads (1 KiB per wave
// instruction). Each W quad is loaded once per step and used for all R rows, so the L2 traffic for W is 1/R of the
// query stream instead of C x larger than it (the single-row form was L2-bound at ~2.3 TB/s). CT classes are
// accumulated at a time in registers (R*CT accumulators); reductions are wave64 shuffles.
template <int CT, int R>
__global__ __launch_bounds__(256) void proto_predict_kernel(
    const float* __restrict__ Q, const float* __restrict__ W, const float* __restrict__ bias, int M,
    int T, int D, int C, float logit_scale, int cosine, float* __restrict__ logits,
    int32_t* __restrict__ argmax) {
    const int task = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + wave) * R;
    if (m0 >= M) return;
    const float* Wt = W + (size_t)task * C * D;
    const float invT = 1.0f / (float)T;
    const bool vec = (D & 3) == 0;
    const float* q[R];
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        row_ok[r] = m0 + r < M;
        q[r] = Q + ((size_t)task * M + (row_ok[r] ? m0 + r : m0)) * T * D;  // clamped: tail rows re-read row m0
    }
    float best[R], qn2[R];
    int best_c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = -INFINITY, best_c[r] = 0, qn2[r] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CT) {
        float dot[R][CT], wn2[CT], qq[R];
#pragma unroll
        for (int j = 0; j < CT; ++j) wn2[j] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            qq[r] = 0.f;
#pragma unroll
            for (int j = 0; j < CT; ++j) dot[r][j] = 0.f;
        }
        if (vec) {
#pragma unroll 4
            for (int d = lane * 4; d < D; d += 256) {  // unrolled: several 1-KiB row segments in flight per wave
                float4 x[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    x[r] = *reinterpret_cast<const float4*>(q[r] + d);
                    for (int t = 1; t < T; ++t) {
                        const float4 y = *reinterpret_cast<const float4*>(q[r] + (size_t)t * D + d);
                        x[r].x += y.x, x[r].y += y.y, x[r].z += y.z, x[r].w += y.w;
                    }
                    if (T > 1) x[r].x *= invT, x[r].y *= invT, x[r].z *= invT, x[r].w *= invT;
                    qq[r] += x[r].x * x[r].x + x[r].y * x[r].y + x[r].z * x[r].z + x[r].w * x[r].w;
                }
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (c0 + j < C) {
                        const float4 w = *reinterpret_cast<const float4*>(Wt + (size_t)(c0 + j) * D + d);
                        if (cosine) wn2[j] += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            dot[r][j] += x[r].x * w.x + x[r].y * w.y + x[r].z * w.z + x[r].w * w.w;
                    }
                }
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                float x[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    x[r] = q[r][d];
                    for (int t = 1; t < T; ++t) x[r] += q[r][(size_t)t * D + d];
                    if (T > 1) x[r] *= invT;
                    qq[r] += x[r] * x[r];
                }
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (c0 + j < C) {
                        const float w = Wt[(size_t)(c0 + j) * D + d];
                        if (cosine) wn2[j] += w * w;
#pragma unroll
                        for (int r = 0; r < R; ++r) dot[r][j] += x[r] * w;
                    }
                }
            }
        }
        if (c0 == 0 && cosine) {
#pragma unroll
            for (int r = 0; r < R; ++r) qn2[r] = wave_sum(qq[r]);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (c0 + j >= C) break;
            const float wn = cosine ? sqrtf(wave_sum(wn2[j])) : 0.f;
            const float bj = cosine ? 0.f : bias[(size_t)task * C + c0 + j];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float v = wave_sum(dot[r][j]);
                if (cosine)
                    v = logit_scale * (v / (fmaxf(sqrtf(qn2[r]), 1e-8f) * fmaxf(wn, 1e-8f)));
                else
                    v = logit_scale * (v + bj);
                if (lane == 0 && row_ok[r]) logits[((size_t)task * M + m0 + r) * C + c0 + j] = v;
                if (v > best[r]) best[r] = v, best_c[r] = c0 + j;
            }
        }
    }
    if (argmax != nullptr && lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row_ok[r]) argmax[(size_t)task * M + m0 + r] = best_c[r];
    }
}

// LDS-staged form for launches with many rows (the batched multi-task launch, and single tasks with >= 64 query rows):
// the block copies the task's weight matrix [C][D] into LDS once, then each of its 4 waves scores 4 query rows at a
// time against it. The one-wave-per-row form above re-reads W from L2 for every row, i.e. C x the query bytes of L2
// traffic (~11 TB/s at 64 tasks x 200 rows x 1280 — L2-bound at 1.7-2.3 TB/s of HBM); here W costs one L2 read per 16
// rows and the query stream is the only HBM traffic. Same per-lane accumulation order as the form above.
template <int CT, int R>
__global__ __launch_bounds__(256) void proto_predict_lds_kernel(
    const float* __restrict__ Q, const float* __restrict__ W, const float* __restrict__ bias, int M, int T, int D, int C,
    float logit_scale, int cosine, float* __restrict__ logits, int32_t* __restrict__ argmax) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [C][D] weights, then [C] norms
    const int task = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* Wt = W + (size_t)task * C * D;
    for (int i = tid * 4; i < C * D; i += 1024) *reinterpret_cast<float4*>(Ws + i) = *reinterpret_cast<const float4*>(Wt + i);
    __syncthreads();
    float* wn = Ws + (size_t)C * D;
    if (cosine) {
        for (int c = wave; c < C; c += 4) {
            float s = 0.f;
            for (int d = lane * 4; d < D; d += 256) {
                const float4 w = *reinterpret_cast<const float4*>(Ws + (size_t)c * D + d);
                s += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
            }
            s = wave_sum(s);
            if (lane == 0) wn[c] = sqrtf(s);
        }
        __syncthreads();
    }
    const int m0 = (blockIdx.x * 4 + wave) * R;
    if (m0 >= M) return;
    const float invT = 1.0f / (float)T;
    const float* q[R];
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        row_ok[r] = m0 + r < M;
        q[r] = Q + ((size_t)task * M + (row_ok[r] ? m0 + r : m0)) * T * D;
    }
    float best[R], qn2[R];
    int best_c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = -INFINITY, best_c[r] = 0, qn2[r] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CT) {
        float dot[R][CT], qq[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            qq[r] = 0.f;
#pragma unroll
            for (int j = 0; j < CT; ++j) dot[r][j] = 0.f;
        }
#pragma unroll 2
        for (int d = lane * 4; d < D; d += 256) {
            float4 x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                x[r] = *reinterpret_cast<const float4*>(q[r] + d);
                for (int t = 1; t < T; ++t) {
                    const float4 y = *reinterpret_cast<const float4*>(q[r] + (size_t)t * D + d);
                    x[r].x += y.x, x[r].y += y.y, x[r].z += y.z, x[r].w += y.w;
                }
                if (T > 1) x[r].x *= invT, x[r].y *= invT, x[r].z *= invT, x[r].w *= invT;
                qq[r] += x[r].x * x[r].x + x[r].y * x[r].y + x[r].z * x[r].z + x[r].w * x[r].w;
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                if (c0 + j < C) {
                    const float4 w = *reinterpret_cast<const float4*>(Ws + (size_t)(c0 + j) * D + d);
#pragma unroll
                    for (int r = 0; r < R; ++r) dot[r][j] += x[r].x * w.x + x[r].y * w.y + x[r].z * w.z + x[r].w * w.w;
                }
            }
        }
        if (c0 == 0 && cosine) {
#pragma unroll
            for (int r = 0; r < R; ++r) qn2[r] = wave_sum(qq[r]);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            if (c0 + j >= C) break;
            const float bj = cosine ? 0.f : bias[(size_t)task * C + c0 + j];
            const float wnj = cosine ? wn[c0 + j] : 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float v = wave_sum(dot[r][j]);
                if (cosine)
                    v = logit_scale * (v / (fmaxf(sqrtf(qn2[r]), 1e-8f) * fmaxf(wnj, 1e-8f)));
                else
                    v = logit_scale * (v + bj);
                if (lane == 0 && row_ok[r]) logits[((size_t)task * M + m0 + r) * C + c0 + j] = v;
                if (v > best[r]) best[r] = v, best_c[r] = c0 + j;
            }
        }
    }
    if (argmax != nullptr && lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row_ok[r]) argmax[(size_t)task * M + m0 + r] = best_c[r];
    }
}

// Streaming form of the LDS-staged kernel for T = 1 and D = 256 * NI (D = 1280: NI 5, D = 512: NI 2, the two extractors):
// every wave REQUESTS all of its query rows (R rows x NI float4 per lane = its whole share of the HBM stream) before the
// block stages the task's weights into LDS, so the two memory latencies overlap instead of adding, and the dot products run
// from registers against LDS after the barrier. The 64-task launch (67 MB) is ~8 us of HBM time: what it can lose is
// exactly such serialised latencies (round 1: W staging -> barrier -> first row load, 2.7 TB/s). Same per-lane accumulation
// order as proto_predict_lds_kernel (d ascending), hence bit-identical logits.
// Block -> (task, row block): the hardware deals consecutive block ids round-robin over the 8 XCDs, each with its own L2. A
// task's weights (C x D floats, 25.6 KB) are staged by every one of its row blocks; dealt round-robin, a task's 13 blocks sat
// on 8 different XCDs and its weights were fetched from memory 8 times (PMC: 79.3 MB of traffic per 64-task launch against
// 67.4 MB of algorithmic bytes, 1.18x). The bijective remap below gives every XCD a contiguous run of logical blocks, so the
// blocks of one task share one L2 and the weights cross the fabric once.
__device__ __forceinline__ int head_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

// (Forcing 8 waves per SIMD - launch_bounds(512, 8): 64 VGPRs, four blocks per CU = 1 024 slots for the 832 blocks of the
// 64-task launch - spills 16 dwords per lane and measured 38.6 us against 19.5 us: the 79-register form stays.)
// LEAN: Euclidean distance, no argmax output - the per-row cosine norms and the running best class leave the register file
// (79 -> <= 72 VGPRs: seven waves per SIMD, so four 7-wave blocks share a CU)
template <int NW, int R, int NI, bool LEAN = false>
__global__ __launch_bounds__(NW * 64) void proto_predict_stream_kernel(
    const float* __restrict__ Q, const float* __restrict__ W, const float* __restrict__ bias, int M, int D, int C,
    float logit_scale, int cosine, float* __restrict__ logits, int32_t* __restrict__ argmax, int blocks_per_task) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [C][D] weights, then [C] norms
    const int logical = head_xcd_remap(blockIdx.x, gridDim.x);
    const int task = logical / blocks_per_task, bx = logical - task * blocks_per_task;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = (bx * NW + wave) * R;
    const bool wave_ok = m0 < M;
    float4 x[R][NI];
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        row_ok[r] = m0 + r < M;
        const float* q = Q + ((size_t)task * M + (row_ok[r] ? m0 + r : (wave_ok ? m0 : 0))) * D + lane * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) x[r][i] = *reinterpret_cast<const float4*>(q + i * 256);  // all in flight at once
    }
    const float* Wt = W + (size_t)task * C * D;
    for (int i = tid * 4; i < C * D; i += NW * 256) *reinterpret_cast<float4*>(Ws + i) = *reinterpret_cast<const float4*>(Wt + i);
    __syncthreads();
    float* wn = Ws + (size_t)C * D;
    if (!LEAN && cosine) {
        for (int c = wave; c < C; c += NW) {
            float s = 0.f;
            for (int d = lane * 4; d < D; d += 256) {
                const float4 w = *reinterpret_cast<const float4*>(Ws + (size_t)c * D + d);
                s += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
            }
            s = wave_sum(s);
            if (lane == 0) wn[c] = sqrtf(s);
        }
        __syncthreads();
    }
    if (!wave_ok) return;
    float best[R], qn2[R];
    int best_c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = -INFINITY, best_c[r] = 0, qn2[r] = 0.f;
    if (!LEAN && cosine) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                qq += x[r][i].x * x[r][i].x + x[r][i].y * x[r][i].y + x[r][i].z * x[r][i].z + x[r][i].w * x[r][i].w;
            qn2[r] = wave_sum(qq);
        }
    }
    // one class at a time (rolled): only NI weight quads are live beside the R x NI query quads. Unrolled over the classes
    // hipcc hoists all C x NI LDS reads - 190-256 VGPRs, one or two waves per SIMD - and a block that computes with
    // nothing in flight then leaves HBM idle (standalone probe tools/head_probe.hip: the same row stream reaches 4.5 TB/s
    // without the weights, 2.5 with them at that occupancy)
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        const float* wrow = Ws + (size_t)c * D + lane * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 w = *reinterpret_cast<const float4*>(wrow + i * 256);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] += x[r][i].x * w.x + x[r][i].y * w.y + x[r][i].z * w.z + x[r][i].w * w.w;
        }
        const float bj = (!LEAN && cosine) ? 0.f : bias[(size_t)task * C + c];
        const float wnj = (!LEAN && cosine) ? wn[c] : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float v = wave_sum(acc[r]);
            if (!LEAN && cosine)
                v = logit_scale * (v / (fmaxf(sqrtf(qn2[r]), 1e-8f) * fmaxf(wnj, 1e-8f)));
            else
                v = logit_scale * (v + bj);
            if (lane == 0 && row_ok[r]) logits[((size_t)task * M + m0 + r) * C + c] = v;
            if (!LEAN && v > best[r]) best[r] = v, best_c[r] = c;
        }
    }
    if (!LEAN && argmax != nullptr && lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row_ok[r]) argmax[(size_t)task * M + m0 + r] = best_c[r];
    }
}

// ---- MeanPooler --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_pool_kernel(const float* __restrict__ x, int N, int T, int D,
                                                        float* __restrict__ out) {
    const size_t total = (size_t)N * D;
    const float invT = 1.0f / (float)T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t n = i / D, d = i % D;
        const float* p = x + n * T * D + d;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += p[(size_t)t * D];
        out[i] = s * invT;
    }
}

// Clip pooling for a video whose clips are sliding windows of its frames (reference data/utils.py:8-28
// attach_frame_history followed by MeanPooler): out[f] = mean_{j<T} x[max(f - T + 1 + j, 0)], summed in clip order, so
// the frame features are computed ONCE per frame instead of once per (frame, window position).
__global__ __launch_bounds__(256) void history_mean_pool_kernel(const float* __restrict__ x, int F, int T, int D,
                                                                float* __restrict__ out) {
    const size_t total = (size_t)F * D;
    const float invT = 1.0f / (float)T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int f = (int)(i / D), d = (int)(i % D);
        float s = 0.f;
        for (int j = 0; j < T; ++j) {
            const int src = f - T + 1 + j;
            s += x[(size_t)(src < 0 ? 0 : src) * D + d];
        }
        out[i] = s * invT;
    }
}

// out[d] = mean_i x[i][d]; one thread per column, rows in ascending order (n is a few hundred)
__global__ __launch_bounds__(64) void set_mean_kernel(const float* __restrict__ x, int n, int D,
                                                      float* __restrict__ out) {
    const int d = blockIdx.x * 64 + threadIdx.x;
    if (d >= D) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += x[(size_t)i * D + d];
    out[d] = s / (float)n;
}

}  // namespace orbit

using namespace orbit;

extern "C" {

int orbit_version(void) { return 100; }

/* Once per device the library is used on: keep freed stream-ordered allocations in the device's default memory pool. The
 * few entry points that take scratch with hipMallocAsync / hipFreeAsync (single-operator test entries, the FiLM generator's
 * backward) otherwise hit a pool whose release threshold is 0: every synchronisation trims it, the next call allocates for
 * real, and the real free that follows synchronises the device under the host's feet. */
int orbit_runtime_init(void) {
    int dev = 0;
    ORBIT_HIP_CHECK(hipGetDevice(&dev));
    static bool done[64] = {false};
    if (dev < 0 || dev >= 64 || done[dev]) return ORBIT_OK;
    hipMemPool_t pool = nullptr;
    ORBIT_HIP_CHECK(hipDeviceGetDefaultMemPool(&pool, dev));
    uint64_t keep = UINT64_MAX;
    ORBIT_HIP_CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    done[dev] = true;
    return ORBIT_OK;
}

int orbit_set_option(const char* name, int value) {
    ORBIT_REQUIRE(name, "set_option: null name");
    Option* o = find_option(name);
    ORBIT_REQUIRE(o != nullptr, "set_option: unknown option '%s'", name);
    if (o->value != value) ++g_option_epoch;
    o->value = value;
    return ORBIT_OK;
}
int orbit_get_option(const char* name) {
    if (!name) return -1;
    return find_option(name) ? get_option(name) : -1;
}
const char* orbit_last_error(void) { return err_buf(); }

int orbit_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_err(ORBIT_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int orbit_proto_configure(const float* feats, const int64_t* labels, const int64_t* class_ids,
                          int n_tasks, int N, int T, int D, int C, float* sums, float* counts,
                          orbit_stream_t stream) {
    ORBIT_REQUIRE(feats && labels && class_ids && sums && counts, "proto_configure: null pointer");
    ORBIT_REQUIRE(n_tasks > 0 && N > 0 && T > 0 && D > 0 && C > 0, "proto_configure: bad sizes");
    ORBIT_REQUIRE(C <= 65535 && n_tasks <= 65535, "proto_configure: C/n_tasks too large");
    dim3 grid(cdiv(D, 256), C, n_tasks);
    // (per-launch event record for bench.py's roofline.families: SURVEY section 8(d): 4 (N D + C D + C) + 8 N bytes per task)
    const int rec = prof_start("head_configure", 2.0 * n_tasks * N * T * D,
                               (double)n_tasks * (4.0 * ((double)N * T * D + (double)C * D + C) + 8.0 * N), (hipStream_t)stream);
    proto_configure_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(feats, labels, class_ids, N, T, D, C,
                                                                  sums, counts);
    prof_stop(rec, (hipStream_t)stream);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_label_set(const int64_t* labels, int N, int64_t* class_ids, int cap, int32_t* count, orbit_stream_t stream) {
    ORBIT_REQUIRE(labels && class_ids && count, "label_set: null pointer");
    ORBIT_REQUIRE(N >= 0 && cap > 0, "label_set: bad sizes");
    const int rec = prof_start("head_label_set", 0.0, 8.0 * N + 8.0 * cap + 4.0, (hipStream_t)stream);
    label_set_kernel<<<1, 64, 0, (hipStream_t)stream>>>(labels, N, class_ids, cap, count);
    prof_stop(rec, (hipStream_t)stream);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_proto_finalize(const float* sums, const float* counts, int n_tasks, int C, int D, int cosine,
                         float* W, float* b, orbit_stream_t stream) {
    ORBIT_REQUIRE(sums && counts && W, "proto_finalize: null pointer");
    ORBIT_REQUIRE(cosine || b, "proto_finalize: euclidean head needs a bias buffer");
    ORBIT_REQUIRE(n_tasks > 0 && D > 0 && C > 0, "proto_finalize: bad sizes");
    dim3 grid(C, n_tasks);
    const int rec = prof_start("head_finalize", 3.0 * n_tasks * C * D, (double)n_tasks * 4.0 * (2.0 * C * D + 2.0 * C),
                               (hipStream_t)stream);
    proto_finalize_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(sums, counts, C, D, cosine, W, b);
    prof_stop(rec, (hipStream_t)stream);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

static int proto_predict_impl(const float* Q, const float* W, const float* b, int n_tasks, int M, int T, int D, int C,
                              float logit_scale, int cosine, float* logits, int32_t* argmax, orbit_stream_t stream);

int orbit_proto_predict(const float* Q, const float* W, const float* b, int n_tasks, int M, int T, int D,
                        int C, float logit_scale, int cosine, float* logits, int32_t* argmax,
                        orbit_stream_t stream) {
    // (per-launch event record for bench.py's roofline.families: SURVEY section 8(d): 4 (M T D + C D + C + M C) bytes per task)
    const int rec = prof_start("head_predict", 2.0 * n_tasks * M * D * C,
                               (double)n_tasks * 4.0 * ((double)M * T * D + (double)C * D + C + (double)M * C), (hipStream_t)stream);
    const int rc = proto_predict_impl(Q, W, b, n_tasks, M, T, D, C, logit_scale, cosine, logits, argmax, stream);
    prof_stop(rec, (hipStream_t)stream);
    return rc;
}

}  // extern "C"

static int proto_predict_impl(const float* Q, const float* W, const float* b, int n_tasks, int M, int T, int D, int C,
                              float logit_scale, int cosine, float* logits, int32_t* argmax, orbit_stream_t stream) {
    ORBIT_REQUIRE(Q && W && logits, "proto_predict: null pointer");
    ORBIT_REQUIRE(cosine || b, "proto_predict: weight and/or bias not set - is the model personalised?");
    ORBIT_REQUIRE(n_tasks > 0 && M > 0 && T > 0 && D > 0 && C > 0, "proto_predict: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // rows per wave: measured on MI355X (64 tasks x 200 x 1280): R = 1 with the row loop unrolled streams faster than
    // R = 4 (more waves in flight beats W reuse: W is L1/L2-resident anyway); the R > 1 forms stay for very wide heads
    const size_t lds = ((size_t)C * D + C) * sizeof(float);
    if ((D & 3) == 0 && lds <= 60 * 1024 && (long)M * n_tasks >= 64) {
        // head_stream (default 1; 0 = the general LDS form below, parity tests): rows requested before the weight staging, one
        // class at a time (T = 1, D = 1280 / 512), 8 waves x 2 rows per block (79 VGPRs: 6 waves per SIMD)
        if (get_option("head_stream") && T == 1 && (D == 1280 || D == 512)) {
#define ORBIT_HEAD_STREAM(NW_, R_, NI_)                                                                                  \
    proto_predict_stream_kernel<NW_, R_, NI_><<<cdiv(M, NW_ * R_) * n_tasks, NW_ * 64, lds, s>>>(                         \
        Q, W, b, M, D, C, logit_scale, cosine, logits, argmax, cdiv(M, NW_ * R_))
            if (D == 1280 && !cosine && argmax == nullptr)
                // the lean instantiation (67 VGPRs): 18.4 against 19.3 us on the 64-task 5-way launch, 24.6 against 26.2 us 10-way
                // (tools/head_roofline.py). A 7-wave form whose 960 blocks are all resident at once (4 x 7 waves per CU) measured
                // 19.5 us: residency is not what bounds this launch, its ~8 us of launch + drain on a 67 MB burst is
                proto_predict_stream_kernel<8, 2, 5, true><<<cdiv(M, 16) * n_tasks, 512, lds, s>>>(
                    Q, W, b, M, D, C, logit_scale, cosine, logits, argmax, cdiv(M, 16));
            else { if (D == 1280) ORBIT_HEAD_STREAM(8, 2, 5); else ORBIT_HEAD_STREAM(8, 2, 2); }
#undef ORBIT_HEAD_STREAM
            ORBIT_LAUNCH_CHECK();
            return ORBIT_OK;
        }
        {
            const dim3 grid(cdiv(M, 16), n_tasks);
            if (C <= 5)
                proto_predict_lds_kernel<5, 4><<<grid, 256, lds, s>>>(Q, W, b, M, T, D, C, logit_scale, cosine, logits, argmax);
            else
                proto_predict_lds_kernel<10, 4><<<grid, 256, lds, s>>>(Q, W, b, M, T, D, C, logit_scale, cosine, logits, argmax);
        }
        ORBIT_LAUNCH_CHECK();
        return ORBIT_OK;
    }
    const long blocks4 = (long)cdiv(M, 16) * n_tasks;
    if (C <= 5) {
        if (blocks4 >= (1L << 30))
            proto_predict_kernel<5, 4><<<dim3(cdiv(M, 16), n_tasks), 256, 0, s>>>(Q, W, b, M, T, D, C, logit_scale, cosine,
                                                                            logits, argmax);
        else
            proto_predict_kernel<5, 1><<<dim3(cdiv(M, 4), n_tasks), 256, 0, s>>>(Q, W, b, M, T, D, C, logit_scale, cosine,
                                                                           logits, argmax);
    } else {
        if (blocks4 >= (1L << 30))
            proto_predict_kernel<10, 2><<<dim3(cdiv(M, 8), n_tasks), 256, 0, s>>>(Q, W, b, M, T, D, C, logit_scale, cosine,
                                                                            logits, argmax);
        else
            proto_predict_kernel<10, 1><<<dim3(cdiv(M, 4), n_tasks), 256, 0, s>>>(Q, W, b, M, T, D, C, logit_scale, cosine,
                                                                            logits, argmax);
    }
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

extern "C" {

int orbit_mean_pool(const float* x, int N, int T, int D, float* out, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && out && N > 0 && T > 0 && D > 0, "mean_pool: bad arguments");
    const size_t total = (size_t)N * D;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    mean_pool_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, N, T, D, out);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_history_mean_pool(const float* x, int F, int T, int D, float* out, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && out && F > 0 && T > 0 && D > 0, "history_mean_pool: bad arguments");
    const size_t total = (size_t)F * D;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    history_mean_pool_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, F, T, D, out);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_set_mean(const float* x, int n, int D, float* out, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && out && n > 0 && D > 0, "set_mean: bad arguments");
    set_mean_kernel<<<cdiv(D, 64), 64, 0, (hipStream_t)stream>>>(x, n, D, out);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

}  // extern "C"
