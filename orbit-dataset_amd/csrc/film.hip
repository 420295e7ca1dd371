// FiLM parameter generator: task embedding z -> per-task BatchNorm (gamma', beta') for every FiLM slot.
//
// Reference: model/feature_adapters.py:36-78 (FilmParameterGenerator.forward) with
// model/mlps.py:52-63 (DenseBlock = Linear -> LayerNorm -> ReLU -> Linear). The reference runs one tiny
// MLP per FiLM tensor (34 for efficientnet_b0, 40 for resnet18) = ~200 launch-bound ATen ops per task; here
// all generators run in ONE grouped launch: block (gen, chunk) recomputes the 64-wide hidden vector
// (4 K MACs) and produces up to 256 outputs, writing straight into the concatenated gamma'/beta'
// vectors that the extractor's BatchNorm folding consumes.
#include <string>
#include <vector>
#include "common.h"

namespace orbit {

struct GenDesc {
    int out, kind, dst;               // outputs, 0=weight/1=bias, offset in film_gamma/film_beta
    size_t w1, b1, ln_w, ln_b;        // offsets (floats) into the parameter pool
    size_t w2, b2, reg, init;
};

constexpr int FILM_MAX_HID = 256;

__global__ __launch_bounds__(256) void filmgen_kernel(const GenDesc* __restrict__ gens,
                                                      const float* __restrict__ pool,
                                                      const float* __restrict__ z, int z_dim, int hid,
                                                      float* __restrict__ film_gamma,
                                                      float* __restrict__ film_beta) {
    __shared__ float zs[FILM_MAX_HID];
    __shared__ float hs[FILM_MAX_HID];
    __shared__ float red[2];
    const GenDesc g = gens[blockIdx.x];
    const int o = blockIdx.y * 256 + threadIdx.x;
    if (blockIdx.y * 256 >= g.out) return;  // uniform per block
    const int t = threadIdx.x;
    if (t < z_dim) zs[t] = z[t];
    __syncthreads();
    float h = 0.f;
    if (t < hid) {
        const float* w = pool + g.w1 + (size_t)t * z_dim;
        for (int k = 0; k < z_dim; ++k) h = fmaf(w[k], zs[k], h);
        h += pool[g.b1 + t];
        hs[t] = h;
    }
    __syncthreads();
    if (t == 0) {  // LayerNorm statistics (biased variance, eps 1e-5), fixed order
        float m = 0.f;
        for (int k = 0; k < hid; ++k) m += hs[k];
        m /= (float)hid;
        float v = 0.f;
        for (int k = 0; k < hid; ++k) v += (hs[k] - m) * (hs[k] - m);
        v /= (float)hid;
        red[0] = m;
        red[1] = 1.0f / sqrtf(v + 1e-5f);
    }
    __syncthreads();
    if (t < hid) {
        const float n = (h - red[0]) * red[1] * pool[g.ln_w + t] + pool[g.ln_b + t];
        hs[t] = fmaxf(n, 0.f);
    }
    __syncthreads();
    if (o < g.out) {
        const float* w = pool + g.w2 + (size_t)o * hid;
        float s = 0.f;
        for (int k = 0; k < hid; ++k) s = fmaf(w[k], hs[k], s);
        s += pool[g.b2 + o];
        const float r = pool[g.reg + o], init = pool[g.init + o];
        if (g.kind == 0)
            film_gamma[g.dst + o] = init * (s * r + 1.0f);
        else
            film_beta[g.dst + o] = init + s * r;
    }
}

// l2 = sum over all generators of sum(reg^2); single block, fixed reduction order
__global__ __launch_bounds__(256) void film_l2_kernel(const GenDesc* __restrict__ gens, int n_gen,
                                                      const float* __restrict__ pool, float* __restrict__ l2) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = 0; i < n_gen; ++i) {
        const GenDesc g = gens[i];
        for (int o = threadIdx.x; o < g.out; o += 256) {
            const float r = pool[g.reg + o];
            s = fmaf(r, r, s);
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) l2[0] = part[0];
}

// Backward of filmgen_kernel + film_l2_kernel for one generator per block. Recomputes the hidden activations (4 K
// MACs) instead of taping them. grads has the layout of the parameter pool (w1 b1 ln_w ln_b w2 b2 reg; the `init`
// slot is not a parameter and is left untouched); dz_partial[gen][z_dim] is summed by filmgen_dz_kernel in generator
// order. Reference: autograd through model/feature_adapters.py:66-78 and model/mlps.py:52-63.
__global__ __launch_bounds__(256) void filmgen_bwd_kernel(const GenDesc* __restrict__ gens,
                                                          const float* __restrict__ pool, const float* __restrict__ z,
                                                          int z_dim, int hid, const float* __restrict__ dgamma,
                                                          const float* __restrict__ dbeta,
                                                          const float* __restrict__ dl2, float* __restrict__ grads,
                                                          float* __restrict__ dz_partial) {
    extern __shared__ float ds[];  // [max_out] gradient w.r.t. the second Linear's outputs
    __shared__ float zs[FILM_MAX_HID], xh[FILM_MAX_HID], hr[FILM_MAX_HID], dxh[FILM_MAX_HID], dhs[FILM_MAX_HID];
    __shared__ float red[2];
    const GenDesc g = gens[blockIdx.x];
    const int t = threadIdx.x;
    if (t < z_dim) zs[t] = z[t];
    __syncthreads();
    float h = 0.f;
    if (t < hid) {
        const float* w = pool + g.w1 + (size_t)t * z_dim;
        for (int k = 0; k < z_dim; ++k) h = fmaf(w[k], zs[k], h);
        h += pool[g.b1 + t];
        xh[t] = h;
    }
    __syncthreads();
    if (t == 0) {
        float m = 0.f;
        for (int k = 0; k < hid; ++k) m += xh[k];
        m /= (float)hid;
        float v = 0.f;
        for (int k = 0; k < hid; ++k) v += (xh[k] - m) * (xh[k] - m);
        v /= (float)hid;
        red[0] = m;
        red[1] = 1.0f / sqrtf(v + 1e-5f);
    }
    __syncthreads();
    const float rstd = red[1];
    float xhat = 0.f;
    if (t < hid) {
        xhat = (h - red[0]) * rstd;
        hr[t] = fmaxf(xhat * pool[g.ln_w + t] + pool[g.ln_b + t], 0.f);
    }
    __syncthreads();
    if (t < hid) xh[t] = xhat;
    const float l2g = dl2 ? 2.0f * dl2[0] : 0.f;
    for (int o = t; o < g.out; o += 256) {
        const float* w = pool + g.w2 + (size_t)o * hid;
        float s = 0.f;
        for (int k = 0; k < hid; ++k) s = fmaf(w[k], hr[k], s);
        s += pool[g.b2 + o];
        const float r = pool[g.reg + o], init = pool[g.init + o];
        const float dout = g.kind == 0 ? dgamma[g.dst + o] * init : dbeta[g.dst + o];
        const float dso = dout * r;
        grads[g.reg + o] = dout * s + l2g * r;
        grads[g.b2 + o] = dso;
        ds[o] = dso;
        float* gw = grads + g.w2 + (size_t)o * hid;
        for (int k = 0; k < hid; ++k) gw[k] = dso * hr[k];
    }
    __syncthreads();
    if (t < hid) {
        float dhr = 0.f;
        for (int o = 0; o < g.out; ++o) dhr = fmaf(ds[o], pool[g.w2 + (size_t)o * hid + t], dhr);
        const float dn = hr[t] > 0.f ? dhr : 0.f;
        grads[g.ln_w + t] = dn * xh[t];
        grads[g.ln_b + t] = dn;
        dxh[t] = dn * pool[g.ln_w + t];
    }
    __syncthreads();
    if (t == 0) {
        float m1 = 0.f, m2 = 0.f;
        for (int k = 0; k < hid; ++k) m1 += dxh[k], m2 += dxh[k] * xh[k];
        red[0] = m1 / (float)hid, red[1] = m2 / (float)hid;
    }
    __syncthreads();
    if (t < hid) {
        const float dh = rstd * (dxh[t] - red[0] - xh[t] * red[1]);
        grads[g.b1 + t] = dh;
        dhs[t] = dh;
        float* gw = grads + g.w1 + (size_t)t * z_dim;
        for (int k = 0; k < z_dim; ++k) gw[k] = dh * zs[k];
    }
    __syncthreads();
    if (t < z_dim) {
        float s = 0.f;
        for (int j = 0; j < hid; ++j) s = fmaf(dhs[j], pool[g.w1 + (size_t)j * z_dim + t], s);
        dz_partial[(size_t)blockIdx.x * z_dim + t] = s;
    }
}

__global__ __launch_bounds__(256) void filmgen_dz_kernel(const float* __restrict__ dz_partial, int n_gen, int z_dim,
                                                         float* __restrict__ dz) {
    const int k = threadIdx.x;
    if (k >= z_dim) return;
    float s = 0.f;
    for (int i = 0; i < n_gen; ++i) s += dz_partial[(size_t)i * z_dim + k];
    dz[k] = s;
}

}  // namespace orbit

using namespace orbit;

struct orbit_filmgen {
    int n_gen = 0, z_dim = 0, hid = 0, max_out = 0;
    std::vector<GenDesc> gens;
    size_t pool_floats = 0;
    float* d_pool = nullptr;
    GenDesc* d_gens = nullptr;
    // orbit_filmgen_load_all_async: device table of source pointers + (offset, numel) of each destination
    const float** d_src = nullptr;
    size_t* d_dst_meta = nullptr;
    std::vector<const float*> h_src;
    float* d_dzp = nullptr;  // [n_gen][z_dim]: per-generator dz of orbit_filmgen_backward (calls are stream-ordered by the caller)
};

// one kernel copies every generator tensor into the pool: grid (chunks, tensors)
__global__ __launch_bounds__(256) void filmgen_gather_kernel(const float* const* __restrict__ src,
                                                             const size_t* __restrict__ meta, float* __restrict__ pool) {
    const float* s_ = src[blockIdx.y];
    float* d = pool + meta[2 * blockIdx.y];
    const size_t n = meta[2 * blockIdx.y + 1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s_[i];
}

extern "C" {

int orbit_filmgen_create(int n_gen, int z_dim, int hidden, const int* out_size, const int* kind,
                         const int* dst_offset, orbit_filmgen_t** out) {
    ORBIT_REQUIRE(out && out_size && kind && dst_offset, "filmgen_create: null pointer");
    ORBIT_REQUIRE(n_gen > 0 && z_dim > 0 && z_dim <= FILM_MAX_HID && hidden > 0 && hidden <= FILM_MAX_HID,
                  "filmgen_create: bad sizes (n_gen=%d z=%d hid=%d)", n_gen, z_dim, hidden);
    orbit_filmgen* g = new orbit_filmgen();
    g->n_gen = n_gen, g->z_dim = z_dim, g->hid = hidden;
    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 3) / 4 * 4;
        return o;
    };
    for (int i = 0; i < n_gen; ++i) {
        GenDesc d;
        d.out = out_size[i], d.kind = kind[i], d.dst = dst_offset[i];
        d.w1 = take((size_t)hidden * z_dim), d.b1 = take(hidden), d.ln_w = take(hidden), d.ln_b = take(hidden);
        d.w2 = take((size_t)d.out * hidden), d.b2 = take(d.out), d.reg = take(d.out), d.init = take(d.out);
        g->gens.push_back(d);
        if (d.out > g->max_out) g->max_out = d.out;
    }
    g->pool_floats = off;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->d_pool), off * sizeof(float));
    if (e == hipSuccess) e = hipMemset(g->d_pool, 0, off * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->d_gens), sizeof(GenDesc) * n_gen);
    if (e == hipSuccess) e = hipMemcpy(g->d_gens, g->gens.data(), sizeof(GenDesc) * n_gen, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(g->d_pool);
        (void)hipFree(g->d_gens);
        delete g;
        return set_err(ORBIT_ERR_HIP, "filmgen_create: %s", hipGetErrorString(e));
    }
    *out = g;
    return ORBIT_OK;
}

void orbit_filmgen_destroy(orbit_filmgen_t* g) {
    if (!g) return;
    (void)hipFree(g->d_pool);
    (void)hipFree(g->d_gens);
    (void)hipFree(g->d_src);
    (void)hipFree(g->d_dst_meta);
    (void)hipFree(g->d_dzp);
    delete g;
}

int orbit_filmgen_load(orbit_filmgen_t* g, int gen, const char* tensor, const float* data, size_t numel) {
    ORBIT_REQUIRE(g && tensor && data, "filmgen_load: null pointer");
    ORBIT_REQUIRE(gen >= 0 && gen < g->n_gen, "filmgen_load: generator %d out of range", gen);
    const GenDesc& d = g->gens[gen];
    const std::string t(tensor);
    size_t off = 0, expect = 0;
    if (t == "w1") off = d.w1, expect = (size_t)g->hid * g->z_dim;
    else if (t == "b1") off = d.b1, expect = g->hid;
    else if (t == "ln_w") off = d.ln_w, expect = g->hid;
    else if (t == "ln_b") off = d.ln_b, expect = g->hid;
    else if (t == "w2") off = d.w2, expect = (size_t)d.out * g->hid;
    else if (t == "b2") off = d.b2, expect = d.out;
    else if (t == "reg") off = d.reg, expect = d.out;
    else if (t == "init") off = d.init, expect = d.out;
    else return set_err(ORBIT_ERR_ARG, "filmgen_load: unknown tensor '%s'", tensor);
    ORBIT_REQUIRE(numel == expect, "filmgen_load: %s of generator %d has %zu elements, expected %zu", tensor,
                  gen, numel, expect);
    ORBIT_HIP_CHECK(hipMemcpy(g->d_pool + off, data, numel * sizeof(float), hipMemcpyDefault));
    return ORBIT_OK;
}

int orbit_filmgen_load_all_async(orbit_filmgen_t* g, const float* const* device_ptrs, int n, orbit_stream_t stream) {
    ORBIT_REQUIRE(g && device_ptrs, "filmgen_load_all_async: null pointer");
    ORBIT_REQUIRE(n == 8 * g->n_gen, "filmgen_load_all_async: %d pointers for %d generators x 8 tensors", n, g->n_gen);
    hipStream_t s = (hipStream_t)stream;
    if (!g->d_src) {
        ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->d_src), n * sizeof(float*)));
        ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->d_dst_meta), 2 * n * sizeof(size_t)));
        std::vector<size_t> meta(2 * n);
        for (int i = 0; i < g->n_gen; ++i) {
            const GenDesc& d = g->gens[i];
            const size_t off[8] = {d.w1, d.b1, d.ln_w, d.ln_b, d.w2, d.b2, d.reg, d.init};
            const size_t num[8] = {(size_t)g->hid * g->z_dim, (size_t)g->hid, (size_t)g->hid, (size_t)g->hid,
                                   (size_t)d.out * g->hid, (size_t)d.out, (size_t)d.out, (size_t)d.out};
            for (int t = 0; t < 8; ++t) meta[2 * (8 * i + t)] = off[t], meta[2 * (8 * i + t) + 1] = num[t];
        }
        ORBIT_HIP_CHECK(hipMemcpy(g->d_dst_meta, meta.data(), meta.size() * sizeof(size_t), hipMemcpyHostToDevice));
    }
    bool same = (int)g->h_src.size() == n;
    for (int i = 0; same && i < n; ++i) same = g->h_src[i] == device_ptrs[i];
    if (!same) {  // the tensors moved: refresh the pointer table (an earlier gather on this stream may still read it)
        for (int i = 0; i < n; ++i) ORBIT_REQUIRE(device_ptrs[i], "filmgen_load_all_async: null tensor %d", i);
        g->h_src.assign(device_ptrs, device_ptrs + n);
        ORBIT_HIP_CHECK(hipStreamSynchronize(s));
        ORBIT_HIP_CHECK(hipMemcpy(g->d_src, g->h_src.data(), n * sizeof(float*), hipMemcpyHostToDevice));
    }
    filmgen_gather_kernel<<<dim3(4, n), 256, 0, s>>>(g->d_src, g->d_dst_meta, g->d_pool);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_filmgen_forward(orbit_filmgen_t* g, const float* z, float* film_gamma, float* film_beta, float* l2,
                          orbit_stream_t stream) {
    ORBIT_REQUIRE(g && z && film_gamma && film_beta, "filmgen_forward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(g->n_gen, cdiv(g->max_out, 256));
    filmgen_kernel<<<grid, 256, 0, s>>>(g->d_gens, g->d_pool, z, g->z_dim, g->hid, film_gamma, film_beta);
    ORBIT_LAUNCH_CHECK();
    if (l2) {
        film_l2_kernel<<<1, 256, 0, s>>>(g->d_gens, g->n_gen, g->d_pool, l2);
        ORBIT_LAUNCH_CHECK();
    }
    return ORBIT_OK;
}

size_t orbit_filmgen_grad_floats(const orbit_filmgen_t* g) { return g ? g->pool_floats : 0; }

size_t orbit_filmgen_param_offset(const orbit_filmgen_t* g, int gen, const char* tensor) {
    if (!g || !tensor || gen < 0 || gen >= g->n_gen) return (size_t)-1;
    const GenDesc& d = g->gens[gen];
    const std::string t(tensor);
    if (t == "w1") return d.w1;
    if (t == "b1") return d.b1;
    if (t == "ln_w") return d.ln_w;
    if (t == "ln_b") return d.ln_b;
    if (t == "w2") return d.w2;
    if (t == "b2") return d.b2;
    if (t == "reg") return d.reg;
    return (size_t)-1;
}

int orbit_filmgen_backward(orbit_filmgen_t* g, const float* z, const float* dfilm_gamma, const float* dfilm_beta,
                           const float* dl2, float* grads, float* dz, orbit_stream_t stream) {
    ORBIT_REQUIRE(g && z && dfilm_gamma && dfilm_beta && grads, "filmgen_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // (a plan-owned buffer: hipMallocAsync / hipFreeAsync here made the host wait for the stream once per LITE step - 23 ms of
    // the 31 ms CNAPs step were spent inside this call)
    if (g->d_dzp == nullptr) ORBIT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&g->d_dzp), (size_t)g->n_gen * g->z_dim * sizeof(float)));
    float* dzp = g->d_dzp;
    filmgen_bwd_kernel<<<g->n_gen, 256, (size_t)g->max_out * sizeof(float), s>>>(g->d_gens, g->d_pool, z, g->z_dim, g->hid,
                                                                                dfilm_gamma, dfilm_beta, dl2, grads, dzp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && dz) {
        filmgen_dz_kernel<<<1, 256, 0, s>>>(dzp, g->n_gen, g->z_dim, dz);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return set_err(ORBIT_ERR_HIP, "filmgen_backward: %s", hipGetErrorString(e));
    return ORBIT_OK;
}

}  // extern "C"
