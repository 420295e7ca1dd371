// Plan / parameter structures of the network runtime, shared by the forward runtime (extractor.hip) and the
// training runtime (extractor_train.hip).
#pragma once
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "common.h"

namespace orbit {

struct Param {
    std::string key;
    size_t numel = 0, off = 0;  // offset (floats) into the parameter pool
    bool loaded = false;
};

struct BNDesc {        // host side
    int gamma, beta, mean, var;  // param indices
    int conv_bias;               // param index or -1
    int C;
    float eps;
    int film_slot;               // -1 if not FiLM-modulated
    int film_off;                // offset inside film_gamma / film_beta
    size_t fold_off;             // offset of this layer's scale/shift in the fold arrays
    std::string name;
};

struct BNDev {         // device descriptor for the fold kernel
    size_t gamma, beta, mean, var, conv_bias;  // pool offsets (conv_bias = SIZE_MAX if none)
    size_t fold_off;
    int C, film_off;                           // film_off < 0: not modulated
    float eps;
};

enum OpKind { OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_AVGPOOL, OP_SE, OP_MBFRONT };

// One weight re-layout of a plan (orbit_extractor_finalize / the dgrad filters of the training runtime). A LITE step repacks
// every filter after the optimizer moved it: ~80 + 32 launches of ~4 us each per step as separate kernels; all jobs of a
// plan run as ONE launch (blockIdx.y = job, grid-stride over the job's destination elements).
struct PackJob {
    const float* src;
    float* dst;
    int kind;  // 0 conv OIHW -> [cout_pad][KT] (vector mode), 1 same in stem mode, 2 depthwise [C][1][K][K] -> [K][K][C],
               // 3 transpose [rows][cols] -> [cols][rows], 4 dgrad filter (rotated, channels swapped) -> packed,
               // 5 pointwise filter [Cout][Cin] -> MFMA-fragment order [tile16][chunk16][lane][4] (csrc/pw_rgemm.hip; KT = Cin / 16)
    int Cin, Cout, KH, KW, cin_pad, KT, cout_pad;
    unsigned total;
};
static __global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < j.total; i += gridDim.x * 256) {
        float v = 0.f;
        if (j.kind == 0 || j.kind == 1) {
            const int n = (int)(i / (unsigned)j.KT), k = (int)(i % (unsigned)j.KT);
            const int per = j.kind == 1 ? j.Cin : j.cin_pad;
            const int tap = k / per, ci = k % per;
            if (n < j.Cout && tap < j.KH * j.KW && ci < j.Cin)
                v = j.src[(((size_t)n * j.Cin + ci) * j.KH + tap / j.KW) * j.KW + tap % j.KW];
        } else if (j.kind == 2) {  // Cin = channels, KH = K
            const int c = (int)(i % (unsigned)j.Cin), tap = (int)(i / (unsigned)j.Cin);
            v = j.src[(size_t)c * j.KH * j.KH + tap];
        } else if (j.kind == 3) {  // Cin = rows, Cout = cols of the source
            const int r = (int)(i / (unsigned)j.Cout), c = (int)(i % (unsigned)j.Cout);
            j.dst[(size_t)c * j.Cin + r] = j.src[i];
            continue;
        } else if (j.kind == 5) {
            const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
            const unsigned tc = i >> 8;
            const int c = (int)(tc % (unsigned)j.KT), tile = (int)(tc / (unsigned)j.KT);
            const int n = tile * 16 + (lane & 15), k = c * 16 + 4 * (lane >> 4) + e;
            if (n < j.Cout) v = j.src[(size_t)n * j.Cin + k];
        } else {  // dgrad: n = ci (output channel of the dgrad conv), k = (tap', co)
            const int n = (int)(i / (unsigned)j.KT), k = (int)(i % (unsigned)j.KT);
            const int tap = k / j.cin_pad, co = k % j.cin_pad;
            if (n < j.Cin && tap < j.KH * j.KW && co < j.Cout) {
                const int kh = j.KH - 1 - tap / j.KW, kw = j.KW - 1 - tap % j.KW;
                v = j.src[(((size_t)co * j.Cin + n) * j.KH + kh) * j.KW + kw];
            }
        }
        j.dst[i] = v;
    }
}
// upload a job list (once: the plan's buffers never move) and run it
static inline int run_pack_jobs(const std::vector<PackJob>& jobs, PackJob** d_jobs, hipStream_t s) {
    if (jobs.empty()) return ORBIT_OK;
    if (*d_jobs == nullptr) {
        ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(d_jobs), jobs.size() * sizeof(PackJob)));
        ORBIT_HIP_CHECK(hipMemcpy(*d_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    }
    unsigned most = 0;
    for (const PackJob& j : jobs) most = std::max(most, j.total);
    unsigned gx = (most + 255) / 256;
    if (gx > 1024) gx = 1024;
    pack_jobs_kernel<<<dim3(gx, (unsigned)jobs.size()), 256, 0, s>>>(*d_jobs);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

struct Op {
    OpKind kind;
    int in = -1, out = -1, res = -1;  // buffer ids: -1 frames, 0..2 activations, 100 feats, 101 pooled, 102 gate
    int H = 0, W = 0, Cin = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad_t = 0, pad_l = 0, Ho = 0, Wo = 0;
    int act = ORBIT_ACT_NONE, pool2 = 0, x_nchw = 0, use_gate = 0;
    int weight = -1, bias = -1, bn = -1;  // param / BN indices
    size_t packed_off = 0;                // into the packed-weight pool
    size_t frag_off = SIZE_MAX;           // OP_CONV: the filter in MFMA-fragment order (csrc/pw_rgemm.hip) or SIZE_MAX
    int se_w1 = -1, se_b1 = -1, se_w2 = -1, se_b2 = -1, R = 0;
    // OP_MBFRONT in its stem form (csrc/mbconv_rows.hip): frame size, stem padding, packed [mid][32] stem filter
    bool stem = false;
    int stem_h = 0, stem_w = 0, stem_pt = 0, stem_pl = 0;
    size_t packed_off2 = 0;
    int se_chunks = 0, se_hw = 0;  // squeeze-excite pooling partials produced by the preceding depthwise conv
    int pool_partial = 0;          // depthwise: also emit the pooling partials
    int weight2 = -1, bn2 = -1;    // OP_MBFRONT: depthwise weight (packed at packed_off) and its BatchNorm
    bool rows = false;             // OP_MBFRONT served by the row-streaming kernel (csrc/mbconv_rows.hip)
    int pool_k = 0, pool_pad = 0;
};

// folds eval-mode BatchNorm (+ per-task FiLM gamma/beta, + the bias of the preceding convolution) into scale/shift;
// static: each runtime translation unit launches its own copy (no relocatable device code needed)
static __global__ __launch_bounds__(256) void bn_fold_all_kernel(const BNDev* __restrict__ descs,
                                                          const float* __restrict__ pool,
                                                          const float* __restrict__ film_gamma,
                                                          const float* __restrict__ film_beta,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
    const BNDev d = descs[blockIdx.x];
    for (int c = blockIdx.y * 256 + threadIdx.x; c < d.C; c += gridDim.y * 256) {
        const bool film = film_gamma != nullptr && d.film_off >= 0;
        const float g = film ? film_gamma[d.film_off + c] : pool[d.gamma + c];
        const float b = film ? film_beta[d.film_off + c] : pool[d.beta + c];
        const float sc = g / sqrtf(pool[d.var + c] + d.eps);
        const float cb = d.conv_bias != (size_t)-1 ? pool[d.conv_bias + c] : 0.f;
        scale[d.fold_off + c] = sc;
        shift[d.fold_off + c] = b + (cb - pool[d.mean + c]) * sc;
    }
}


}  // namespace orbit

struct orbit_extractor;
namespace orbit {
void extractor_train_invalidate(const orbit_extractor* fe);  // parameters changed: training-side repacks are stale
void extractor_train_release(const orbit_extractor* fe);     // plan is being destroyed
}  // namespace orbit

using namespace orbit;  // internal header: only included by the two runtime translation units

struct orbit_extractor {
    std::string name;
    int H = 0, W = 0, out_size = 0;
    std::vector<Param> params;
    std::map<std::string, int> index;
    std::vector<BNDesc> bns;
    std::vector<Op> ops;
    std::vector<int> film_slots;  // BN indices in module-traversal order
    int film_size = 0;
    size_t pool_floats = 0, packed_floats = 0, fold_floats = 0;
    size_t buf_elems[3] = {0, 0, 0};  // per-frame element counts of the rotating activation buffers
    int max_se_c = 0;
    size_t max_partial = 0;  // floats per frame of the SE pooling-partial buffer
    double macs = 0;
    float* d_pool = nullptr;
    // batched upload (orbit_extractor_load_all_async): device table of source pointers + its host shadow, and the static
    // table of (pool offset, numel) per parameter
    const float** d_src = nullptr;
    std::vector<const float*> h_src;
    size_t* d_dst_meta = nullptr;  // [n][2] = offset, numel
    float* d_packed = nullptr;
    std::vector<PackJob> pack_jobs;   // filter re-layouts of orbit_extractor_finalize (built at the first call)
    PackJob* d_pack_jobs = nullptr;
    int pack_jobs_bk = -1;            // the conv_bk option the job list was built under (it fixes the packed geometry)
    float* d_fold = nullptr;  // static (non-FiLM) scale | shift
    BNDev* d_bn = nullptr;
    std::vector<BNDev> bn_dev;  // host copy of the fold descriptors
    bool finalized = false;

    // HIP-graph cache: one instantiated graph per distinct (pointers, batch, stream) tuple of forward(). A forward is
    // 25-90 dependent launches; replaying them as one graph launch takes the host out of the loop (on a slow or busy
    // host the eager launch sequence, not the GPU, bounded small workloads).
    struct GraphKey {
        const void *frames, *gamma, *beta, *feats, *ws, *stream;
        int B;
        int epoch;  // option_epoch() at capture: runtime options choose kernels
        bool operator==(const GraphKey& o) const {
            return frames == o.frames && gamma == o.gamma && beta == o.beta && feats == o.feats && ws == o.ws &&
                   stream == o.stream && B == o.B && epoch == o.epoch;
        }
    };
    struct GraphEntry {
        GraphKey key;
        hipGraphExec_t exec = nullptr;  // nullptr: seen once (ran eagerly), capture on the next sight
        unsigned long stamp = 0;
    };
    std::vector<GraphEntry> graphs;
    unsigned long graph_clock = 0;
    double eager_us_per_launch = 0.0;  // running average of the HOST cost of one eager kernel launch (option graph=2)
    int eager_samples = 0;
    hipStream_t cap_stream = nullptr;  // private non-default stream used only to CAPTURE (the legacy default stream,
                                       // torch's default, cannot be captured); graphs are launched on the caller's
    void clear_graphs() {
        for (GraphEntry& g : graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
    }

    // Graph cache of the TRAINING entry points (orbit_extractor_train_forward / orbit_extractor_backward): one LITE step
    // of efficientnet_b0 is ~4 000 dependent launches, and the host, not the GPU, bounded it (39 of 41 ms enqueuing).
    // Keyed by every pointer and scalar that a launch of the sequence bakes in; the plan's own buffers (parameter pool,
    // packed filters, dgrad filters) are allocated once and never move, so these graphs survive parameter updates (the
    // kernels read the CURRENT contents) and are only dropped with the plan. Callers whose allocator hands back the same
    // addresses step after step (torch's caching allocator in a steady-state training loop) replay; others stay eager.
    struct TrainGraphKey {
        const void* p[10];
        long v[5];  // v[4] = option_epoch() (set by run_train_graphed)
        bool operator==(const TrainGraphKey& o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
    };
    struct TrainGraphEntry {
        TrainGraphKey key;
        hipGraphExec_t exec = nullptr;  // nullptr: seen once (ran eagerly); captured on the next sight
        bool dead = false;              // capture failed for this key: stay eager
        unsigned long stamp = 0;
    };
    std::vector<TrainGraphEntry> train_graphs;
    long train_graph_replays = 0, train_graph_eager = 0;
    void clear_train_graphs() {
        for (TrainGraphEntry& g : train_graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        train_graphs.clear();
    }
    template <class F>
    int run_train_graphed(TrainGraphKey key, hipStream_t s, F&& run) {
        // 0 (default) = never, 1 = replay from the third sight of a key on. Opt-in: measured on MI355X the LITE step of
        // efficientnet_b0 is bound by the GPU side of its ~1 260 short kernels (42 ms of kernel time per step), so replay
        // only frees the host (36 -> 26 ms of enqueue time per step) without shortening the step (41.3 vs 41.4 ms)
        const int opt = get_option("train_graph");
        if (opt == 0 || conv_prof_enabled()) {
            ++train_graph_eager;
            return run(s);
        }
        key.v[4] = option_epoch();
        TrainGraphEntry* hit = nullptr;
        for (auto& g : train_graphs)
            if (g.key == key) hit = &g;
        if (hit == nullptr) {
            if (train_graphs.size() >= 64) {  // evict the least recently used entry
                size_t lru = 0;
                for (size_t i = 1; i < train_graphs.size(); ++i)
                    if (train_graphs[i].stamp < train_graphs[lru].stamp) lru = i;
                if (train_graphs[lru].exec) (void)hipGraphExecDestroy(train_graphs[lru].exec);
                train_graphs.erase(train_graphs.begin() + lru);
            }
            TrainGraphEntry e;
            e.key = key, e.stamp = ++graph_clock;
            train_graphs.push_back(e);
            ++train_graph_eager;
            return run(s);  // first sight: eager (also performs one-time kernel attribute set-up)
        }
        hit->stamp = ++graph_clock;
        if (hit->dead) {
            ++train_graph_eager;
            return run(s);
        }
        if (hit->exec == nullptr) {
            hipGraph_t graph = nullptr;
            if (cap_stream == nullptr && hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                hit->dead = true;
                return run(s);
            }
            if (hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();
                hit->dead = true;
                return run(s);
            }
            const int rc = run(cap_stream);
            const hipError_t ce = hipStreamEndCapture(cap_stream, &graph);
            hipGraphExec_t exec = nullptr;
            if (rc == ORBIT_OK && ce == hipSuccess && graph != nullptr &&
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec != nullptr) {
                hit->exec = exec;
            } else {
                (void)hipGetLastError();
                hit->dead = true;
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (rc != ORBIT_OK) return rc;
            if (hit->dead) {
                ++train_graph_eager;
                return run(s);
            }
        }
        if (hipGraphLaunch(hit->exec, s) != hipSuccess) {
            (void)hipGetLastError();
            hit->dead = true;
            ++train_graph_eager;
            return run(s);
        }
        ++train_graph_replays;
        return ORBIT_OK;
    }

    // device buffers are created on first use so that a plan can be built and inspected (state_dict keys,
    // FiLM slots, workspace size, MACs) on a host without a GPU
    int ensure_device() {
        if (d_pool) return ORBIT_OK;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_pool), pool_floats * sizeof(float));
        if (e == hipSuccess) e = hipMemset(d_pool, 0, pool_floats * sizeof(float));
        if (e == hipSuccess)
            e = hipMalloc(reinterpret_cast<void**>(&d_packed), std::max<size_t>(packed_floats, 4) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_fold), 2 * fold_floats * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_bn), bn_dev.size() * sizeof(BNDev));
        if (e == hipSuccess)
            e = hipMemcpy(d_bn, bn_dev.data(), bn_dev.size() * sizeof(BNDev), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d_pool), (void)hipFree(d_packed), (void)hipFree(d_fold), (void)hipFree(d_bn);
            d_pool = d_packed = d_fold = nullptr, d_bn = nullptr;
            (void)hipGetLastError();
            return set_err(ORBIT_ERR_HIP, "extractor: device allocation failed: %s", hipGetErrorString(e));
        }
        return ORBIT_OK;
    }

    int add_param(const std::string& key, size_t numel) {
        Param p;
        p.key = key, p.numel = numel, p.off = pool_floats;
        pool_floats += (numel + 3) / 4 * 4;
        params.push_back(p);
        index[key] = (int)params.size() - 1;
        return (int)params.size() - 1;
    }
    int add_bn(const std::string& prefix, int C, float eps, bool film, int conv_bias = -1) {
        BNDesc b;
        b.name = prefix;
        b.gamma = add_param(prefix + ".weight", C);
        b.beta = add_param(prefix + ".bias", C);
        b.mean = add_param(prefix + ".running_mean", C);
        b.var = add_param(prefix + ".running_var", C);
        b.conv_bias = conv_bias, b.C = C, b.eps = eps;
        b.film_slot = -1, b.film_off = -1;
        b.fold_off = fold_floats;
        fold_floats += (size_t)(C + 3) / 4 * 4;
        if (film) {
            b.film_slot = (int)film_slots.size();
            b.film_off = film_size;
            film_size += C;
            film_slots.push_back((int)bns.size());
        }
        bns.push_back(b);
        return (int)bns.size() - 1;
    }
    void note_buf(int id, size_t elems) {
        if (id >= 0 && id < 3) buf_elems[id] = std::max(buf_elems[id], elems);
    }
    // dense conv + BN (+act) (+residual) (+gate) (+pool2); returns output dims through Ho/Wo
    void add_conv(const std::string& wkey, int bn, int in, int out, int res, int H_, int W_, int Cin, int Cout,
                  int K, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, int pool2, int x_nchw,
                  int use_gate, int bias = -1) {
        Op o;
        o.kind = OP_CONV, o.in = in, o.out = out, o.res = res;
        o.H = H_, o.W = W_, o.Cin = Cin, o.Cout = Cout, o.KH = K, o.KW = K, o.stride = stride;
        o.pad_t = pad_t, o.pad_l = pad_l, o.Ho = Ho, o.Wo = Wo, o.act = act, o.pool2 = pool2;
        o.x_nchw = x_nchw, o.use_gate = use_gate, o.bn = bn, o.bias = bias;
        o.weight = index.count(wkey) ? index[wkey] : add_param(wkey, (size_t)Cout * Cin * K * K);
        o.packed_off = packed_floats;
        packed_floats += conv_packed_floats(Cin, Cout, K, K, x_nchw);
        // the fragment-ordered copy of a pointwise filter only where the register GEMM will be asked for it (csrc/pw_rgemm.hip:
        // conv_rgemm = 1 one layer class, = 2 every supported conv) - it doubles the packed bytes and the pack work of a layer
        ConvDesc shape;
        shape.H = H_, shape.W = W_, shape.Cin = Cin, shape.Cout = Cout;
        const int rg = get_option("conv_rgemm");
        if (stride == 1 && pad_t == 0 && pad_l == 0 && !pool2 && conv_frag_floats(Cin, Cout, K, K, x_nchw) &&
            (rg == 2 || (rg == 1 && pw_rgemm_preferred(shape)))) {
            packed_floats = (packed_floats + 63) & ~(size_t)63;
            o.frag_off = packed_floats;
            packed_floats += conv_frag_floats(Cin, Cout, K, K, x_nchw);
        }
        const int oh = pool2 ? Ho / 2 : Ho, ow = pool2 ? Wo / 2 : Wo;
        note_buf(out, (size_t)oh * ow * Cout);
        macs += (double)(pool2 ? oh * 2 : Ho) * (pool2 ? ow * 2 : Wo) * Cout * Cin * K * K;
        ops.push_back(o);
    }
};

