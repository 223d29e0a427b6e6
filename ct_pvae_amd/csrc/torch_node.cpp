// torch_node.cpp -- the drop-in API's autograd nodes for the training layout, in C++.
//
// RotateVae: project_tf_fast(x [B][X][Y][1], theta, pad, dim=2, integrate_vae=True) -> [B][A][P][1] and its backward are
// ONE torch::autograd::Function whose forward and backward each make one call into the C ABI of include/ctpvae_radon.h
// (ctpvae_rotate_fwd_planned_f32 / ctpvae_rotate_bwd_planned_scaled_f32).
// RotateLogLik: calculate_log_prob_M_given_R of a planned geometry with a fixed pnm, optionally at an angle subset of a
// dense plan (ctvae/helper_functions.py:336-368): forward = the one-launch projection + log-likelihood that also stores
// d lp / d sino, backward = the projector's backward with the upstream per-object factor applied in its store.
// Nothing is computed here: this file only removes the Python that torch.autograd.Function.apply and a Python backward
// cost per call (~6 us of ~27, ~16 of ~52).  Host code only -- no HIP headers: the library is bound with dlopen / dlsym
// (the pointer types are taken from the C header, so a changed signature does not compile) and the raw stream arrives as an
// integer from Python (the backward runs on the stream the forward ran on, as the autograd engine arranges).  Built by
// __graft_entry__.build() with torch.utils.cpp_extension (g++), in-tree.
#include <dlfcn.h>
#include <torch/extension.h>

#include "ctpvae_radon.h"

namespace {

using fwd_fn = decltype(&ctpvae_rotate_fwd_planned_f32);
using bwd_fn = decltype(&ctpvae_rotate_bwd_planned_scaled_f32);
using err_fn = decltype(&ctpvae_last_error);
using fwd_lik_fn = decltype(&ctpvae_rotate_fwd_planned_loglik_f32);
using fwd_lik_sel_fn = decltype(&ctpvae_rotate_fwd_planned_loglik_sel_f32);
using bwd_seg_fn = decltype(&ctpvae_rotate_bwd_scaled_f32);
using bwd_sel_fn = decltype(&ctpvae_rotate_bwd_sel_scaled_f32);
using fwd_compact_fn = decltype(&ctpvae_rotate_fwd_compact_f32);
using bwd_psel_fn = decltype(&ctpvae_rotate_bwd_planned_sel_scaled_f32);
using bwd_step_fn = decltype(&ctpvae_rotate_bwd_stepped_scaled_f32);
using abi_fn = decltype(&ctpvae_abi_version);
fwd_fn g_fwd = nullptr;
bwd_fn g_bwd = nullptr;
err_fn g_err = nullptr;
fwd_lik_fn g_fwd_lik = nullptr;
fwd_lik_sel_fn g_fwd_lik_sel = nullptr;
bwd_seg_fn g_bwd_seg = nullptr;
bwd_sel_fn g_bwd_sel = nullptr;
fwd_compact_fn g_fwd_compact = nullptr;
bwd_psel_fn g_bwd_psel = nullptr;
bwd_step_fn g_bwd_step = nullptr;

// The ABI this node was COMPILED against (the header's macro) must be the ABI of the library it binds at run time: the
// entry points are resolved by name only, so a node left over from an older build would otherwise call them with an old
// argument list.  Returns the library's version; the Python side (ct_pvae_amd/_lib.py torch_node()) compares and falls back
// to the Python nodes on a mismatch.
int64_t compiled_abi() { return CTPVAE_ABI_VERSION; }

void bind(const std::string &lib_path)
{
    void *h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    TORCH_CHECK(h != nullptr, "cannot load ", lib_path, ": ", dlerror());
    abi_fn abi = (abi_fn)dlsym(h, "ctpvae_abi_version");
    TORCH_CHECK(abi != nullptr && abi() == CTPVAE_ABI_VERSION, "ct_pvae_amd torch node was built for ABI ", CTPVAE_ABI_VERSION,
                " but ", lib_path, " has ABI ", abi ? abi() : -1, ": rebuild the node (__graft_entry__.build())");
    g_fwd = (fwd_fn)dlsym(h, "ctpvae_rotate_fwd_planned_f32");
    g_bwd = (bwd_fn)dlsym(h, "ctpvae_rotate_bwd_planned_scaled_f32");
    g_err = (err_fn)dlsym(h, "ctpvae_last_error");
    g_fwd_lik = (fwd_lik_fn)dlsym(h, "ctpvae_rotate_fwd_planned_loglik_f32");
    g_fwd_lik_sel = (fwd_lik_sel_fn)dlsym(h, "ctpvae_rotate_fwd_planned_loglik_sel_f32");
    g_bwd_seg = (bwd_seg_fn)dlsym(h, "ctpvae_rotate_bwd_scaled_f32");
    g_bwd_sel = (bwd_sel_fn)dlsym(h, "ctpvae_rotate_bwd_sel_scaled_f32");
    g_fwd_compact = (fwd_compact_fn)dlsym(h, "ctpvae_rotate_fwd_compact_f32");
    g_bwd_psel = (bwd_psel_fn)dlsym(h, "ctpvae_rotate_bwd_planned_sel_scaled_f32");
    g_bwd_step = (bwd_step_fn)dlsym(h, "ctpvae_rotate_bwd_stepped_scaled_f32");
    TORCH_CHECK(g_fwd && g_bwd && g_err && g_fwd_lik && g_fwd_lik_sel && g_bwd_seg && g_bwd_sel && g_fwd_compact && g_bwd_psel && g_bwd_step,
                lib_path,
                " does not export the planned projector entry points");
}

// the binding's error convention (ct_pvae_amd/_lib.py check()): CTPVAE_EINVAL (-1) -> ValueError, anything else -> RuntimeError
void check(int rc, const char *what)
{
    TORCH_CHECK_VALUE(rc != -1, what, ": ", g_err());
    TORCH_CHECK(rc >= 0, what, ": ", g_err());
}

struct RotateVae : public torch::autograd::Function<RotateVae> {
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &x4, const at::Tensor &fwd_plan,
                              const at::Tensor &bwd_plan, int64_t H, int64_t W, int64_t PH, int64_t PW, int64_t A, int64_t stream,
                              int64_t compact)
    {
        TORCH_CHECK(g_fwd != nullptr, "ct_pvae_amd torch node: bind() was not called");
        TORCH_CHECK(x4.is_cuda() && x4.dim() == 4 && x4.size(1) == H && x4.size(2) == W && x4.size(3) == 1 && x4.size(0) > 0 &&
                        x4.scalar_type() == at::kFloat && x4.is_contiguous() && x4.device() == fwd_plan.device(),
                    "project_tf_fast: expected a contiguous float32 [B][", H, "][", W, "][1] tensor on ", fwd_plan.device());
        const int64_t S = x4.size(0);
        at::Tensor out = at::empty({S, A, PW, 1}, x4.options());
        const int rc = compact ? g_fwd_compact(x4.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A,
                                               fwd_plan.data_ptr(), nullptr, 0, 0, nullptr, nullptr, 0, nullptr, 0.0f,
                                               out.data_ptr<float>(), nullptr, nullptr, nullptr, nullptr, (void *)stream)
                               : g_fwd(x4.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A, fwd_plan.data_ptr(),
                                       out.data_ptr<float>(), (void *)stream);
        check(rc, "rotate_fwd");
        ctx->saved_data["bwd_plan"] = bwd_plan;
        ctx->saved_data["geo"] = std::vector<int64_t>{H, W, PH, PW, A, stream};
        return out;
    }

    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::tensor_list grads)
    {
        const auto geo = ctx->saved_data["geo"].toIntVector();
        const at::Tensor bwd_plan = ctx->saved_data["bwd_plan"].toTensor();
        at::Tensor g = grads[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        const int64_t S = g.size(0), H = geo[0], W = geo[1], PH = geo[2], PW = geo[3], A = geo[4];
        at::Tensor gimg = at::empty({S, H, W, 1}, g.options());
        const int rc = g_bwd(g.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A, bwd_plan.data_ptr(), nullptr, 0,
                             gimg.data_ptr<float>(), (void *)geo[5]);
        check(rc, "rotate_bwd");
        torch::autograd::tensor_list out(10);
        out[0] = gimg;
        return out;
    }
};

// geo = {H, W, PH, PW, A_plan, py, px, backward plan kind (0: none -- segment kernel, 1: the dense plan, 2: the bwd4 plan of
// an angle subset), dense_inputs, stream, compact forward plan}; angles: int32 [n] on the device or undefined
struct RotateLogLik : public torch::autograd::Function<RotateLogLik> {
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &x4, const at::Tensor &fwd_plan,
                              const at::Tensor &bwd_plan, const at::Tensor &Tinv8, const at::Tensor &mask, const at::Tensor &meas,
                              const at::Tensor &pnm, const c10::optional<at::Tensor> &angles, double eps, std::vector<int64_t> geo)
    {
        TORCH_CHECK(g_fwd_lik != nullptr, "ct_pvae_amd torch node: bind() was not called");
        const int64_t H = geo[0], W = geo[1], PH = geo[2], PW = geo[3], A = geo[4], dense = geo[8], stream = geo[9], compact = geo[10];
        const bool sel = angles.has_value() && angles->defined();
        const int64_t n = sel ? angles->numel() : A, n_in = (sel && !dense) ? n : A;
        auto f32c = [&](const at::Tensor &t) { return t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous() && t.device() == x4.device(); };
        TORCH_CHECK(f32c(x4) && x4.dim() == 4 && x4.size(0) > 0 && x4.size(1) == H && x4.size(2) == W && x4.size(3) == 1 &&
                        x4.device() == fwd_plan.device(),
                    "calculate_log_prob_M_given_R: expected a contiguous float32 [B][", H, "][", W, "][1] tensor on ", fwd_plan.device());
        const int64_t S = x4.size(0);
        TORCH_CHECK_VALUE(f32c(mask) && f32c(meas) && f32c(pnm) && pnm.numel() == 1 && mask.dim() == 2 && mask.size(0) == S &&
                              mask.size(1) == n_in && meas.dim() == 3 && meas.size(0) == S && meas.size(1) == n_in && meas.size(2) == PW,
                          "need contiguous float32 mask [", S, "][", n_in, "], proj_sample [", S, "][", n_in, "][", PW,
                          "] and a one-element pnm on ", x4.device());
        const bool host_idx = sel && !angles->is_cuda();   // host-resident subsets ride the launch arguments (compact plan only)
        TORCH_CHECK_VALUE(!sel || (angles->scalar_type() == at::kInt && angles->is_contiguous() && n > 0 &&
                                   (angles->is_cuda() || (compact && geo[7] == 2 && n <= 256))),
                          "angles_i must be a non-empty contiguous int32 vector, on the device or (compact plans) in host memory");
        at::Tensor sino = at::empty({S, n, PW}, x4.options());
        at::Tensor lp = at::empty({S, n, PW, 1}, x4.options());
        at::Tensor dlp = at::empty({S, n, PW}, x4.options());
        int rc;
        if (compact)
            rc = g_fwd_compact(x4.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A, fwd_plan.data_ptr(),
                               sel ? angles->data_ptr<int>() : nullptr, sel ? (int)n : 0, host_idx ? 1 : 0, mask.data_ptr<float>(),
                               meas.data_ptr<float>(), (int)dense, pnm.data_ptr<float>(), (float)eps, sino.data_ptr<float>(),
                               lp.data_ptr<float>(), dlp.data_ptr<float>(), nullptr, nullptr, (void *)stream);
        else if (sel)
            rc = g_fwd_lik_sel(x4.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A, fwd_plan.data_ptr(),
                               angles->data_ptr<int>(), (int)n, mask.data_ptr<float>(), meas.data_ptr<float>(), (int)dense,
                               pnm.data_ptr<float>(), (float)eps, sino.data_ptr<float>(), lp.data_ptr<float>(),
                               dlp.data_ptr<float>(), (void *)stream);
        else
            rc = g_fwd_lik(x4.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A, fwd_plan.data_ptr(),
                           mask.data_ptr<float>(), meas.data_ptr<float>(), pnm.data_ptr<float>(), (float)eps,
                           sino.data_ptr<float>(), lp.data_ptr<float>(), dlp.data_ptr<float>(), (void *)stream);
        check(rc, "rotate_fwd_planned_loglik");
        ctx->saved_data["dlp"] = dlp;
        ctx->saved_data["bwd_plan"] = bwd_plan;
        ctx->saved_data["Tinv8"] = Tinv8;
        if (sel) ctx->saved_data["angles"] = *angles;
        ctx->saved_data["geo"] = geo;
        return lp;
    }

    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::tensor_list grads)
    {
        const auto geo = ctx->saved_data["geo"].toIntVector();
        const int64_t H = geo[0], W = geo[1], PH = geo[2], PW = geo[3], A = geo[4], py = geo[5], px = geo[6], use_plan = geo[7];
        void *stream = (void *)geo[9];
        at::Tensor dlp = ctx->saved_data["dlp"].toTensor();
        const bool sel = ctx->saved_data.count("angles") != 0;
        at::Tensor g = grads[0];                       // [S][n][PW][1]
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        const int64_t S = dlp.size(0), n = dlp.size(1);
        const float *scale = nullptr;
        long long stride = 0;
        at::Tensor cot = dlp;
        if (g.stride(1) == 0 && g.stride(2) == 0) {    // the gradient of a per-object sum: one factor per slice
            scale = g.data_ptr<float>();
            stride = g.stride(0);
        } else {
            cot = (g.squeeze(-1) * dlp).contiguous();
        }
        at::Tensor gimg = at::empty({S, H, W, 1}, dlp.options());
        int rc;
        if (sel && use_plan == 2) {                    // the bwd4 plan: a planned backward that selects angles
            const at::Tensor ai = ctx->saved_data["angles"].toTensor();
            rc = g_bwd_psel(cot.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A,
                            ctx->saved_data["bwd_plan"].toTensor().data_ptr(), ai.data_ptr<int>(), (int)n, ai.is_cuda() ? 0 : 1,
                            scale, stride,
                            gimg.data_ptr<float>(), stream);
        } else if (sel) {
            const at::Tensor ai = ctx->saved_data["angles"].toTensor();
            rc = g_bwd_sel(cot.data_ptr<float>(), (int)S, (int)A, (int)PH, (int)PW, ctx->saved_data["Tinv8"].toTensor().data_ptr<float>(),
                           ai.data_ptr<int>(), (int)n, (int)H, (int)W, (int)py, (int)px, scale, stride, gimg.data_ptr<float>(), stream);
        } else if (use_plan == 3) {                    // the step plan: large batches (the entry point picks its kernel)
            rc = g_bwd_step(cot.data_ptr<float>(), (int)S, (int)A, (int)PH, (int)PW, ctx->saved_data["Tinv8"].toTensor().data_ptr<float>(),
                            (int)H, (int)W, (int)py, (int)px, ctx->saved_data["bwd_plan"].toTensor().data_ptr(), scale, stride,
                            gimg.data_ptr<float>(), stream);
        } else if (use_plan) {
            rc = g_bwd(cot.data_ptr<float>(), (int)S, (int)H, (int)W, (int)PH, (int)PW, (int)A,
                       ctx->saved_data["bwd_plan"].toTensor().data_ptr(), scale, stride, gimg.data_ptr<float>(), stream);
        } else {                                       // interp NEAREST (0), mode TF_COMPAT (0): the segment kernel
            rc = g_bwd_seg(cot.data_ptr<float>(), (int)S, (int)A, (int)PH, (int)PW, ctx->saved_data["Tinv8"].toTensor().data_ptr<float>(),
                           0, 0, (int)H, (int)W, (int)py, (int)px, scale, stride, gimg.data_ptr<float>(), stream);
        }
        check(rc, "rotate_bwd");
        torch::autograd::tensor_list out(10);
        out[0] = gimg;
        return out;
    }
};

at::Tensor rotate_loglik(const at::Tensor &x4, const at::Tensor &fwd_plan, const at::Tensor &bwd_plan, const at::Tensor &Tinv8,
                         const at::Tensor &mask, const at::Tensor &meas, const at::Tensor &pnm, const c10::optional<at::Tensor> &angles,
                         double eps, std::vector<int64_t> geo)
{
    TORCH_CHECK(geo.size() == 11, "geo = {H, W, PH, PW, A, py, px, backward_uses_plan, dense_inputs, stream, compact}");
    return RotateLogLik::apply(x4, fwd_plan, bwd_plan, Tinv8, mask, meas, pnm, angles, eps, geo);
}

at::Tensor rotate_vae(const at::Tensor &x4, const at::Tensor &fwd_plan, const at::Tensor &bwd_plan, int64_t H, int64_t W, int64_t PH,
                      int64_t PW, int64_t A, int64_t stream, int64_t compact)
{
    return RotateVae::apply(x4, fwd_plan, bwd_plan, H, W, PH, PW, A, stream, compact);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("compiled_abi", &compiled_abi, "CTPVAE_ABI_VERSION of the header this node was compiled against");
    m.def("bind", &bind, "dlopen libctpvae_radon.so and resolve the planned projector entry points");
    m.def("rotate_loglik", &rotate_loglik,
          "calculate_log_prob_M_given_R for [B][X][Y][1] float32 through a gather plan: projection + log-likelihood in one launch");
    m.def("rotate_vae", &rotate_vae, "project_tf_fast for [B][X][Y][1] float32 through a gather plan (differentiable)");
}
