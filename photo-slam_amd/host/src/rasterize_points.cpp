// rasterize_points.cpp -- the torch <-> kernel boundary (replaces src/rasterize_points.cu:28-214 of the reference;
// third_party/simple-knn/spatial.cu:15-26 is spatial.cpp) on top of the C-ABI of libgsr_hip.so.
// LibTorch supplies device memory and the current stream only.
#include "rasterize_points.h"

#include <string>

#include "../../../include/gsr.h"

#ifndef GSR_HOST_NO_HIP
#include <c10/hip/HIPStream.h>
#endif

namespace {

// Sizes of a megabyte and more are rounded up to the next eighth of the power of two below them (at most 12.5 % more): the binning
// buffer's size follows the instance count, which creeps from step to step of a training run, and a caching allocator answers every
// size it has not seen with a fresh hipMalloc (up to 45 ms for a gigabyte on the pool's boxes) while the blocks it cannot reuse pile up
// (EXPERIMENTS R6.8).  Bucketed, the requests recur and are served from the cache.  The buffers are opaque bytes: a larger one is
// the same to every consumer.
size_t bucket_bytes(size_t bytes)
{
	if (bytes < (size_t(1) << 20)) return bytes;
	size_t p = size_t(1) << 20;
	while ((p << 1) <= bytes) p <<= 1;
	const size_t step = p >> 3;
	return (bytes + step - 1) / step * step;
}

// resizeFunctional, src/rasterize_points.cu:28-34, as a plain C callback (sizes bucketed, see above)
char* resize_tensor(void* ctx, size_t bytes)
{
	auto* t = static_cast<torch::Tensor*>(ctx);
	t->resize_({static_cast<int64_t>(bucket_bytes(bytes))});
	return reinterpret_cast<char*>(t->data_ptr());
}

// a RasterWorkspace buffer (rasterize_points.h): kept when large enough, replaced by one with 50 % headroom otherwise (a new
// allocation, not resize_: the old content is of no use and would be copied)
char* grow_tensor(void* ctx, size_t bytes)
{
	auto* t = static_cast<torch::Tensor*>(ctx);
	if (static_cast<size_t>(t->numel()) < bytes) {
		const auto opts = t->options();
		*t = torch::Tensor();   // (released first: the two never have to coexist)
		*t = torch::empty({static_cast<int64_t>(bytes + bytes / 2)}, opts);
	}
	return reinterpret_cast<char*>(t->data_ptr());
}

void* current_stream(const torch::Tensor& t)
{
#ifndef GSR_HOST_NO_HIP
	if (t.is_cuda()) return c10::hip::getCurrentHIPStream(t.device().index()).stream();
#endif
	return nullptr;
}

// contiguous fp32 pointer, nullptr for an empty tensor (the reference's "absent optional")
struct F32 {
	torch::Tensor keep;
	const float* ptr = nullptr;
	explicit F32(const torch::Tensor& t)
	{
		if (t.defined() && t.numel() != 0) {
			keep = t.contiguous();
			ptr = keep.data_ptr<float>();
		}
	}
};

void check(int status, const char* where)
{
	if (status == GSR_OK) return;
	std::string msg = std::string(where) + ": " + gsr_strerror(status);
	if (status == GSR_ERR_HIP) msg += std::string(" [") + gsr_last_hip_error_string() + "]";
	throw std::runtime_error(msg);
}

// ShAdamStep -> gsr_sh_adam (+ gsr_sh_adam_lazy when row_step is defined); `lazy` must outlive the call that gets `adam`
void fill_sh_adam(const ShAdamStep& s, float* param, gsr_sh_adam& adam, gsr_sh_adam_lazy& lazy)
{
	adam = gsr_sh_adam{};
	adam.param = param;
	adam.exp_avg = s.exp_avg.data_ptr<float>();
	adam.exp_avg_sq = s.exp_avg_sq.data_ptr<float>();
	adam.lr = s.lr; adam.lr_tail = s.lr_tail;
	adam.beta1 = s.beta1; adam.beta2 = s.beta2; adam.eps = s.eps;
	adam.step = s.step;
	adam.no_side_stream = s.no_side_stream ? 1 : 0;
	adam.lazy_slice_late = s.lazy_slice_late ? 1 : 0;
	adam.side_blocks = s.side_blocks;
	if (s.row_step.defined()) {
		if (s.row_step.scalar_type() != torch::kInt32 || !s.row_step.is_contiguous() || s.row_step.numel() != s.exp_avg.size(0) ||
		    s.row_step.device() != s.exp_avg.device())
			throw std::runtime_error("sh_adam.row_step must be a contiguous int32 [P] tensor on the device of the moments");
		lazy = gsr_sh_adam_lazy{};
		lazy.row_step = s.row_step.data_ptr<int>();
		lazy.window = s.window;
		for (size_t k = 0; k < GSR_SH_LAZY_WINDOW; k++) {
			lazy.lr_past[k] = k < s.lr_past.size() ? s.lr_past[k] : 0.0;
			lazy.lr_tail_past[k] = k < s.lr_tail_past.size() ? s.lr_tail_past[k] : 0.0;
		}
		adam.lazy = &lazy;
	}
}

}  // namespace

// the reference's exact parameter lists (include/rasterize_points.h:18-37, :39-60): same mangled names
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansCUDA(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered)
{
	return RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
	                              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
	                              prefiltered, /*raw_params=*/0);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                               const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer)
{
	return RasterizeGaussiansBackwardCUDA(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
	                                      viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
	                                      geomBuffer, R, binningBuffer, imageBuffer, /*raw_params=*/0, torch::Tensor(),
	                                      ShAdamStep(), std::vector<torch::Tensor>());
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansCUDA(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered, const int raw_params)
{
	return RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
	                              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
	                              prefiltered, raw_params, ShAdamStep());
}

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansCUDA(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered, const int raw_params, const ShAdamStep& sh_adam, RasterWorkspace* workspace)
{
	if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
		AT_ERROR("means3D must have dimensions (num_points, 3)");
	}
	const int P = static_cast<int>(means3D.size(0));
	const int H = image_height, W = image_width;
	auto float_opts = means3D.options().dtype(torch::kFloat32);
	// torch::full(0) in the reference (rasterize_points.cu:68-69); gsr_forward writes every pixel and every radius itself,
	// so only the P == 0 no-op needs the zeros
	torch::Tensor out_color = P != 0 ? torch::empty({3, H, W}, float_opts) : torch::zeros({3, H, W}, float_opts);
	torch::Tensor radii = torch::empty({P}, means3D.options().dtype(torch::kInt32));
	auto byte_opts = means3D.options().dtype(torch::kByte);
	torch::Tensor geomBuffer = torch::empty({0}, byte_opts);
	torch::Tensor binningBuffer = torch::empty({0}, byte_opts);
	torch::Tensor imgBuffer = torch::empty({0}, byte_opts);
	// (a persistent workspace: its buffers are used in place and returned; a buffer of another device or type starts over)
	torch::Tensor* bufs[3] = {&geomBuffer, &binningBuffer, &imgBuffer};
	if (workspace) {
		torch::Tensor* ws[3] = {&workspace->geom, &workspace->binning, &workspace->img};
		for (int i = 0; i < 3; i++) {
			if (!ws[i]->defined() || ws[i]->device() != means3D.device() || ws[i]->scalar_type() != torch::kByte) *ws[i] = torch::empty({0}, byte_opts);
			bufs[i] = ws[i];
		}
	}
	const gsr_alloc_fn take = workspace ? grow_tensor : resize_tensor;

	int rendered = 0;
	if (P != 0) {
		F32 bg(background), m3(means3D), col(colors), op(opacity), sc(scales), rot(rotations), cov(cov3D_precomp),
		    view(viewmatrix), proj(projmatrix), shs(sh), cam(campos);
		gsr_forward_args a{};
		a.P = P;
		a.D = degree;
		a.M = (sh.defined() && sh.numel() != 0) ? static_cast<int>(sh.size(1)) : 0;
		a.background = bg.ptr;
		a.width = W;
		a.height = H;
		a.means3D = m3.ptr;
		a.shs = shs.ptr;
		a.colors_precomp = col.ptr;
		a.opacities = op.ptr;
		a.scales = sc.ptr;
		a.scale_modifier = scale_modifier;
		a.rotations = rot.ptr;
		a.cov3D_precomp = cov.ptr;
		a.viewmatrix = view.ptr;
		a.projmatrix = proj.ptr;
		a.cam_pos = cam.ptr;
		a.tan_fovx = tan_fovx;
		a.tan_fovy = tan_fovy;
		a.prefiltered = prefiltered ? 1 : 0;
		a.raw_params = raw_params;
		a.out_color = out_color.data_ptr<float>();
		a.radii = radii.data_ptr<int>();
		gsr_sh_adam adam{};
		gsr_sh_adam_lazy lazy{};
		if (sh_adam.row_step.defined()) {   // lazy SH Adam: the forward pass brings visible rows up to date (in place)
			if (!sh_adam.exp_avg.defined() || !sh.defined() || sh.scalar_type() != torch::kFloat32 || !sh.is_contiguous() ||
			    !sh_adam.exp_avg.is_contiguous() || !sh_adam.exp_avg_sq.is_contiguous() || sh_adam.exp_avg.sizes() != sh.sizes() ||
			    sh_adam.exp_avg_sq.sizes() != sh.sizes())
				throw std::runtime_error("lazy sh_adam needs contiguous float32 sh and moments of one shape");
			fill_sh_adam(sh_adam, const_cast<float*>(a.shs), adam, lazy);
			a.sh_adam = &adam;
		}
		check(gsr_forward(&a, take, bufs[0], take, bufs[1], take, bufs[2], current_stream(means3D), &rendered), "RasterizeGaussiansCUDA");
	}
	return std::make_tuple(rendered, out_color, radii, *bufs[0], *bufs[1], *bufs[2]);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                               const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                               const int raw_params, const torch::Tensor& dL_dcolor_view, const ShAdamStep& sh_adam,
                               const std::vector<torch::Tensor>& view_stats)
{
	return RasterizeGaussiansBackwardCUDA(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
	                                      viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
	                                      geomBuffer, R, binningBuffer, imageBuffer, raw_params, dL_dcolor_view, sh_adam, view_stats,
	                                      GeomAdamStep());
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                               const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                               const int raw_params, const torch::Tensor& dL_dcolor_view, const ShAdamStep& sh_adam,
                               const std::vector<torch::Tensor>& view_stats, const GeomAdamStep& geom_adam)
{
	const int P = static_cast<int>(means3D.size(0));
	const int H = static_cast<int>(dL_dout_color.size(1));
	const int W = static_cast<int>(dL_dout_color.size(2));
	const int M = (sh.defined() && sh.numel() != 0) ? static_cast<int>(sh.size(1)) : 0;
	auto o = means3D.options().dtype(torch::kFloat32);
	const bool has_sh = M != 0, has_scales = scales.defined() && scales.numel() != 0;
	// the reference zero-fills all nine (src/rasterize_points.cu:149-157); gsr_backward writes every
	// element of every output it is given, so only the outputs it is NOT given need zeros
	// the gradients of the four small parameter tensors are slices of ONE buffer (rotation first: its float4 stores need
	// the 16-byte alignment), so that a data-parallel driver reduces them over the ranks with a single collective; the
	// buffer itself is not kept: autograd adopts a gradient only if nothing else references it
	// the fused geometry step (GeomAdamStep): the four gradients are not computed (dL_dmeans3D stays as scratch)
	const bool geom = !geom_adam.param.empty();
	if (geom) {
		if (geom_adam.param.size() != 4 || geom_adam.exp_avg.size() != 4 || geom_adam.exp_avg_sq.size() != 4 || geom_adam.lr.size() != 4 ||
		    geom_adam.step.size() != 4 || !has_scales || dL_dcolor_view.defined())
			throw std::runtime_error("geom_adam needs four tensors (xyz, opacity, scaling, rotation) with moments, learning rates and steps, scales / rotations, and no dL_dcolor_view");
		for (size_t i = 0; i < 4; i++)
			for (const auto* t : {&geom_adam.param[i], &geom_adam.exp_avg[i], &geom_adam.exp_avg_sq[i]})
				if (!t->defined() || t->scalar_type() != torch::kFloat32 || !t->is_contiguous() || t->device() != means3D.device() ||
				    t->sizes() != geom_adam.param[i].sizes() || t->size(0) != P)
					throw std::runtime_error("geom_adam tensors must be contiguous float32 [num_points, ...] on the device of means3D");
	}
	const bool slim = geom_adam.training_outputs_only;
	if (slim && !has_scales) throw std::runtime_error("training_outputs_only needs scales / rotations (dL_dcov3D is not written)");
	torch::Tensor dL_drotations, dL_dmeans3D, dL_dscales, dL_dopacity;
	if (geom) {
		dL_dmeans3D = torch::empty({P, 3}, o);   // scratch between the two backward kernels
	} else {
		const int64_t n = P;
		torch::Tensor flat = torch::empty({11 * n}, o);
		dL_drotations = flat.narrow(0, 0, 4 * n).view({n, 4});
		dL_dmeans3D = flat.narrow(0, 4 * n, 3 * n).view({n, 3});
		dL_dscales = flat.narrow(0, 7 * n, 3 * n).view({n, 3});
		dL_dopacity = flat.narrow(0, 10 * n, n).view({n, 1});
	}
	torch::Tensor dL_dmeans2D, dL_dcov3D;
	if (!slim) {
		dL_dmeans2D = torch::empty({P, 3}, o);
		dL_dcov3D = torch::empty({P, 6}, o);
	}
	torch::Tensor dL_dcolors = torch::empty({P, 3}, o);
	const bool factored = dL_dcolor_view.defined();
	if (factored && (!has_sh || dL_dcolor_view.dim() != 2 || dL_dcolor_view.size(0) != P || dL_dcolor_view.size(1) != 3 ||
	                 dL_dcolor_view.scalar_type() != torch::kFloat32 || !dL_dcolor_view.is_contiguous() ||
	                 dL_dcolor_view.device() != means3D.device()))
		throw std::runtime_error("dL_dcolor_view must be a contiguous float32 (num_points, 3) tensor on the device of means3D, with SHs");
	const bool fused_adam = sh_adam.exp_avg.defined();
	// (sh_adam together with dL_dcolor_view only in its lazy form: backward then runs this step's slice of the rows' catch-up)
	if (fused_adam && ((factored && !sh_adam.row_step.defined()) || !has_sh || !sh.is_contiguous() || sh.scalar_type() != torch::kFloat32 ||
	                   !sh_adam.exp_avg.is_contiguous() || !sh_adam.exp_avg_sq.is_contiguous() ||
	                   sh_adam.exp_avg.sizes() != sh.sizes() || sh_adam.exp_avg_sq.sizes() != sh.sizes() || sh_adam.step < 1))
		throw std::runtime_error("sh_adam needs contiguous float32 sh and moments of one shape, step >= 1, and no dL_dcolor_view");
	torch::Tensor dL_dsh;
	if (!factored && !fused_adam) dL_dsh = has_sh ? torch::empty({P, M, 3}, o) : torch::zeros({P, M, 3}, o);
	if (!has_scales && !geom) {
		dL_dscales.zero_();
		dL_drotations.zero_();
	}

	if (P != 0) {
		F32 bg(background), m3(means3D), col(colors), sc(scales), rot(rotations), cov(cov3D_precomp), view(viewmatrix),
		    proj(projmatrix), shs(sh), cam(campos), dpix(dL_dout_color);
		torch::Tensor radii_c = radii.contiguous(), geom_c = geomBuffer.contiguous(), bin_c = binningBuffer.contiguous(),
		              img_c = imageBuffer.contiguous();
		gsr_backward_args a{};
		a.P = P;
		a.D = degree;
		a.M = M;
		a.R = R;
		a.background = bg.ptr;
		a.width = W;
		a.height = H;
		a.means3D = m3.ptr;
		a.shs = shs.ptr;
		a.colors_precomp = col.ptr;
		a.scales = sc.ptr;
		a.scale_modifier = scale_modifier;
		a.rotations = rot.ptr;
		a.cov3D_precomp = cov.ptr;
		a.viewmatrix = view.ptr;
		a.projmatrix = proj.ptr;
		a.campos = cam.ptr;
		a.tan_fovx = tan_fovx;
		a.tan_fovy = tan_fovy;
		a.radii = radii_c.numel() ? radii_c.data_ptr<int>() : nullptr;
		a.geom_buffer = reinterpret_cast<char*>(geom_c.data_ptr());
		a.binning_buffer = bin_c.numel() ? reinterpret_cast<char*>(bin_c.data_ptr()) : nullptr;
		a.image_buffer = reinterpret_cast<char*>(img_c.data_ptr());
		a.dL_dpix = dpix.ptr;
		a.dL_dmean2D = dL_dmeans2D.defined() ? dL_dmeans2D.data_ptr<float>() : nullptr;
		a.dL_dconic = nullptr;  // internal to the reference's wrapper (rasterize_points.cu:152)
		a.dL_dopacity = dL_dopacity.defined() ? dL_dopacity.data_ptr<float>() : nullptr;
		a.dL_dcolor = dL_dcolors.data_ptr<float>();
		a.dL_dmean3D = dL_dmeans3D.data_ptr<float>();
		a.dL_dcov3D = dL_dcov3D.defined() ? dL_dcov3D.data_ptr<float>() : nullptr;
		a.dL_dsh = (has_sh && dL_dsh.defined()) ? dL_dsh.data_ptr<float>() : nullptr;
		if (!view_stats.empty()) {
			if (view_stats.size() != 3) throw std::runtime_error("view_stats: {xyz_gradient_accum, denom, max_radii2D}");
			for (const auto& t : view_stats)
				if (t.numel() != P || t.scalar_type() != torch::kFloat32 || !t.is_contiguous() || t.device() != means3D.device())
					throw std::runtime_error("view_stats tensors must be contiguous float32 with num_points elements");
			a.stat_grad_accum = view_stats[0].data_ptr<float>();
			a.stat_denom = view_stats[1].data_ptr<float>();
			a.stat_max_radii = view_stats[2].data_ptr<float>();
		}
		gsr_sh_adam adam{};
		gsr_sh_adam_lazy lazy{};
		if (fused_adam) {
			// the caller handed `sh` over for the in-place update (ShAdamStep contract)
			fill_sh_adam(sh_adam, const_cast<float*>(a.shs), adam, lazy);
			a.sh_adam = &adam;
		}
		a.dL_dcolor_view = factored ? dL_dcolor_view.data_ptr<float>() : nullptr;
		a.color_view_ready_stream = factored ? sh_adam.color_view_ready_stream : nullptr;
		if (factored && sh_adam.packed_view.defined()) {
			const auto& m = sh_adam.packed_view;
			if (m.scalar_type() != torch::kInt32 || !m.is_contiguous() || m.device() != means3D.device() ||
			    m.numel() < packedViewWords(P, sh_adam.packed_capacity))
				throw std::runtime_error("RasterizeGaussiansBackwardCUDA: packed_view must be a contiguous int32 message of "
				                         "packedViewWords(P, packed_capacity) words on the device of means3D");
			a.packed_view = reinterpret_cast<uint32_t*>(m.data_ptr<int32_t>());
			a.packed_capacity_rows = static_cast<int>(sh_adam.packed_capacity);
		}
		a.dL_dscale = (has_scales && !geom) ? dL_dscales.data_ptr<float>() : nullptr;
		a.dL_drot = (has_scales && !geom) ? dL_drotations.data_ptr<float>() : nullptr;
		a.raw_params = raw_params;
		gsr_geom_adam ga{};
		if (geom) {
			gsr_adam_tensor* ts[4] = {&ga.xyz, &ga.opacity, &ga.scaling, &ga.rotation};
			for (size_t i = 0; i < 4; i++) {
				ts[i]->param = geom_adam.param[i].data_ptr<float>();
				ts[i]->exp_avg = geom_adam.exp_avg[i].data_ptr<float>();
				ts[i]->exp_avg_sq = geom_adam.exp_avg_sq[i].data_ptr<float>();
				ts[i]->lr = geom_adam.lr[i];
				ts[i]->step = static_cast<int>(geom_adam.step[i]);
			}
			ga.beta1 = geom_adam.beta1; ga.beta2 = geom_adam.beta2; ga.eps = geom_adam.eps;
			a.geom_adam = &ga;
		}
		check(gsr_backward(&a, current_stream(means3D)), "RasterizeGaussiansBackwardCUDA");
	}
	if (geom) dL_dmeans3D = torch::Tensor();   // (was scratch)
	return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
	                       dL_drotations);
}

namespace {
// the gathered views / centres: dim 0 may be strided (both may be slices of one gathered [n_views, P + 1, 3] buffer -- no
// copy is made of them), the inner dimensions must be dense
struct ViewsArgs {
	const float* campos;
	const float* views;
	long long campos_stride, view_stride;
	int n_views;
};
ViewsArgs views_args(const torch::Tensor& means3D, const torch::Tensor& campos_views, const torch::Tensor& dL_dcolor_views)
{
	const int64_t P = means3D.size(0);
	if (dL_dcolor_views.dim() != 3 || dL_dcolor_views.size(1) != P || dL_dcolor_views.size(2) != 3 || campos_views.dim() != 2 ||
	    campos_views.size(0) != dL_dcolor_views.size(0) || campos_views.size(1) != 3)
		throw std::runtime_error("dL_dcolor_views must be (n_views, num_points, 3) and campos_views (n_views, 3)");
	if (dL_dcolor_views.scalar_type() != torch::kFloat32 || campos_views.scalar_type() != torch::kFloat32 ||
	    dL_dcolor_views.device() != means3D.device() || campos_views.device() != means3D.device())
		throw std::runtime_error("dL_dcolor_views and campos_views must be float32 tensors on the device of means3D");
	if ((P && (dL_dcolor_views.stride(1) != 3 || dL_dcolor_views.stride(2) != 1)) || campos_views.stride(1) != 1)
		throw std::runtime_error("dL_dcolor_views / campos_views: only the view dimension may be strided");
	ViewsArgs a;
	a.n_views = static_cast<int>(dL_dcolor_views.size(0));
	a.campos = campos_views.data_ptr<float>();
	a.views = dL_dcolor_views.data_ptr<float>();
	a.campos_stride = a.n_views > 1 ? campos_views.stride(0) : 3;
	a.view_stride = a.n_views > 1 ? dL_dcolor_views.stride(0) : 3 * P;
	return a;
}
}  // namespace

torch::Tensor shGradFromViews(const torch::Tensor& means3D, const torch::Tensor& campos_views,
                              const torch::Tensor& dL_dcolor_views, const int degree, const int M, const float scale)
{
	const int P = static_cast<int>(means3D.size(0));
	const ViewsArgs va = views_args(means3D, campos_views, dL_dcolor_views);
	torch::Tensor out = torch::empty({P, M, 3}, means3D.options().dtype(torch::kFloat32));
	if (P != 0) {
		F32 m3(means3D);
		check(gsr_sh_grad_from_views(P, degree, M, va.n_views, m3.ptr, va.campos, va.campos_stride, va.views, va.view_stride,
		                             scale, out.data_ptr<float>(), current_stream(means3D)),
		      "shGradFromViews");
	}
	return out;
}

void shAdamFromViews(const torch::Tensor& means3D, const torch::Tensor& campos_views, const torch::Tensor& dL_dcolor_views,
                     const int degree, const float scale, torch::Tensor& sh, const ShAdamStep& sh_adam)
{
	const int P = static_cast<int>(means3D.size(0));
	const ViewsArgs va = views_args(means3D, campos_views, dL_dcolor_views);
	if (sh.dim() != 3 || !sh.is_contiguous() || sh.scalar_type() != torch::kFloat32 || !sh_adam.exp_avg.defined() ||
	    !sh_adam.exp_avg.is_contiguous() || !sh_adam.exp_avg_sq.is_contiguous() || sh_adam.exp_avg.sizes() != sh.sizes() ||
	    sh_adam.exp_avg_sq.sizes() != sh.sizes())
		throw std::runtime_error("sh and its moments must be contiguous float32 (num_points, M, 3) tensors");
	if (P == 0) return;
	F32 m3(means3D);
	// sh_adam.row_step (lazy mode, gsr_sh_adam_lazy): rows no view lights are left alone and step later; the caller runs
	// shAdamLazySlice() after the last row range of the step
	gsr_sh_adam adam{};
	gsr_sh_adam_lazy lazy{};
	fill_sh_adam(sh_adam, sh.data_ptr<float>(), adam, lazy);
	check(gsr_sh_adam_from_views(P, degree, static_cast<int>(sh.size(1)), va.n_views, m3.ptr, va.campos, va.campos_stride,
	                             va.views, va.view_stride, scale, sh.data_ptr<float>(), &adam, current_stream(means3D)),
	      "shAdamFromViews");
}

int lastVisibleCount() { return gsr_last_visible_count(); }

int64_t packedViewWords(int64_t P, int64_t capacity) { return static_cast<int64_t>(gsr_packed_view_words(static_cast<int>(P), static_cast<int>(capacity))); }

void packColorView(const torch::Tensor& dL_dcolor_view, const torch::Tensor& campos, int64_t capacity, torch::Tensor& message,
                   torch::Tensor& scratch)
{
	torch::NoGradGuard ng;
	const int P = static_cast<int>(dL_dcolor_view.size(0));
	if (dL_dcolor_view.dim() != 2 || dL_dcolor_view.size(1) != 3 || !dL_dcolor_view.is_contiguous() ||
	    dL_dcolor_view.scalar_type() != torch::kFloat32)
		throw std::runtime_error("packColorView: dL_dcolor_view must be a contiguous float32 (num_points, 3) tensor");
	if (message.scalar_type() != torch::kInt32 || !message.is_contiguous() || message.numel() < packedViewWords(P, capacity) ||
	    message.device() != dL_dcolor_view.device())
		throw std::runtime_error("packColorView: message must be a contiguous int32 tensor of packedViewWords(P, capacity) words");
	if (P == 0) return;
	const int64_t need = static_cast<int64_t>(gsr_pack_scratch_bytes(P));
	if (!scratch.defined() || scratch.numel() < need || scratch.device() != dL_dcolor_view.device())
		scratch = torch::empty({need}, dL_dcolor_view.options().dtype(torch::kByte));
	F32 c(campos);
	check(gsr_pack_color_view(P, dL_dcolor_view.data_ptr<float>(), c.ptr, static_cast<int>(capacity),
	                          reinterpret_cast<uint32_t*>(message.data_ptr<int32_t>()), scratch.data_ptr(), current_stream(dL_dcolor_view)),
	      "packColorView");
}

void packViewPlan(const torch::Tensor& radii, int64_t capacity, torch::Tensor& message, torch::Tensor& scratch)
{
	torch::NoGradGuard ng;
	const int P = static_cast<int>(radii.size(0));
	if (radii.dim() != 1 || !radii.is_contiguous() || radii.scalar_type() != torch::kInt32)
		throw std::runtime_error("packViewPlan: radii must be a contiguous int32 (num_points) tensor");
	if (message.scalar_type() != torch::kInt32 || !message.is_contiguous() || message.numel() < packedViewWords(P, capacity) ||
	    message.device() != radii.device())
		throw std::runtime_error("packViewPlan: message must be a contiguous int32 tensor of packedViewWords(P, capacity) words");
	if (P == 0) return;
	const int64_t need = static_cast<int64_t>(gsr_pack_scratch_bytes(P));
	if (!scratch.defined() || scratch.numel() < need || scratch.device() != radii.device())
		scratch = torch::empty({need}, radii.options().dtype(torch::kByte));
	check(gsr_pack_view_plan(P, radii.data_ptr<int32_t>(), reinterpret_cast<uint32_t*>(message.data_ptr<int32_t>()), scratch.data_ptr(),
	                         current_stream(radii)),
	      "packViewPlan");
}

void checkPackedViews(const torch::Tensor& messages, int64_t msg_stride, int64_t n_views, int64_t P, int64_t capacity)
{
	if (messages.scalar_type() != torch::kInt32 || !messages.is_contiguous() || messages.numel() < (n_views - 1) * msg_stride + 8)
		throw std::runtime_error("checkPackedViews: messages must be a contiguous int32 tensor of n_views messages, msg_stride words apart");
	const int st = gsr_check_packed_views(static_cast<int>(P), static_cast<int>(n_views),
	                                      reinterpret_cast<const uint32_t*>(messages.data_ptr<int32_t>()), msg_stride,
	                                      static_cast<int>(capacity), current_stream(messages));
	if (st == GSR_ERR_INVALID_ARG)
		throw std::runtime_error("checkPackedViews: a gathered message does not describe " + std::to_string(P) + " rows with capacity " +
		                         std::to_string(capacity) + ", or its sender dropped rows (include/gsr.h: message word [3])");
	check(st, "checkPackedViews");
}

torch::Tensor shGradFromPackedViews(const torch::Tensor& means3D, const torch::Tensor& messages, int64_t msg_stride, int64_t n_views,
                                    const int degree, const int M, const float scale)
{
	const int P = static_cast<int>(means3D.size(0));
	torch::Tensor out = torch::empty({P, M, 3}, means3D.options().dtype(torch::kFloat32));
	if (P != 0) {
		F32 m3(means3D);
		check(gsr_sh_grad_from_packed_views(P, degree, M, static_cast<int>(n_views), m3.ptr,
		                                    reinterpret_cast<const uint32_t*>(messages.data_ptr<int32_t>()), msg_stride, scale,
		                                    out.data_ptr<float>(), current_stream(means3D)),
		      "shGradFromPackedViews");
	}
	return out;
}

void shAdamFromPackedViews(const torch::Tensor& means3D, const torch::Tensor& messages, int64_t msg_stride, int64_t n_views,
                           const int degree, const float scale, torch::Tensor& sh, const ShAdamStep& sh_adam)
{
	const int P = static_cast<int>(means3D.size(0));
	if (sh.dim() != 3 || !sh.is_contiguous() || sh.scalar_type() != torch::kFloat32 || !sh_adam.exp_avg.defined() ||
	    !sh_adam.exp_avg.is_contiguous() || !sh_adam.exp_avg_sq.is_contiguous() || sh_adam.exp_avg.sizes() != sh.sizes() ||
	    sh_adam.exp_avg_sq.sizes() != sh.sizes())
		throw std::runtime_error("sh and its moments must be contiguous float32 (num_points, M, 3) tensors");
	if (P == 0) return;
	F32 m3(means3D);
	gsr_sh_adam adam{};
	gsr_sh_adam_lazy lazy{};
	fill_sh_adam(sh_adam, sh.data_ptr<float>(), adam, lazy);
	check(gsr_sh_adam_from_packed_views(P, degree, static_cast<int>(sh.size(1)), static_cast<int>(n_views), m3.ptr,
	                                    reinterpret_cast<const uint32_t*>(messages.data_ptr<int32_t>()), msg_stride, scale,
	                                    sh.data_ptr<float>(), &adam, current_stream(means3D)),
	      "shAdamFromPackedViews");
}

void shAdamFlush(torch::Tensor& sh, const ShAdamStep& sh_adam)
{
	torch::NoGradGuard ng;
	if (!sh_adam.row_step.defined() || !sh_adam.exp_avg.defined() || sh.dim() != 3 || sh.size(1) != 16 || !sh.is_contiguous() ||
	    sh.scalar_type() != torch::kFloat32 || !sh_adam.exp_avg.is_contiguous() || !sh_adam.exp_avg_sq.is_contiguous() ||
	    sh_adam.exp_avg.sizes() != sh.sizes() || sh_adam.exp_avg_sq.sizes() != sh.sizes())
		throw std::runtime_error("shAdamFlush needs a contiguous float32 [P,16,3] tensor, its moments and row_step");
	gsr_sh_adam adam{};
	gsr_sh_adam_lazy lazy{};
	fill_sh_adam(sh_adam, sh.data_ptr<float>(), adam, lazy);
	check(gsr_sh_adam_flush(static_cast<int>(sh.size(0)), &adam, current_stream(sh)), "shAdamFlush");
}

void shAdamLazySlice(torch::Tensor& sh, const ShAdamStep& sh_adam, bool ahead)
{
	torch::NoGradGuard ng;
	if (!sh_adam.row_step.defined() || !sh_adam.exp_avg.defined() || sh.dim() != 3 || sh.size(1) != 16 || !sh.is_contiguous() ||
	    sh.scalar_type() != torch::kFloat32 || !sh_adam.exp_avg.is_contiguous() || !sh_adam.exp_avg_sq.is_contiguous() ||
	    sh_adam.exp_avg.sizes() != sh.sizes() || sh_adam.exp_avg_sq.sizes() != sh.sizes())
		throw std::runtime_error("shAdamLazySlice needs a contiguous float32 [P,16,3] tensor, its moments and row_step");
	gsr_sh_adam adam{};
	gsr_sh_adam_lazy lazy{};
	fill_sh_adam(sh_adam, sh.data_ptr<float>(), adam, lazy);
	check(gsr_sh_adam_lazy_slice(static_cast<int>(sh.size(0)), &adam, ahead ? 1 : 0, current_stream(sh)), "shAdamLazySlice");
}

void adamStepMulti(const std::vector<AdamMultiEntry>& entries, double beta1, double beta2, double eps)
{
	torch::NoGradGuard ng;
	if (entries.empty()) return;
	std::vector<gsr_adam_multi_tensor> ts;
	for (const auto& e : entries) {
		for (const torch::Tensor* t : {&e.param, &e.grad, &e.exp_avg, &e.exp_avg_sq})
			if (!t->defined() || t->scalar_type() != torch::kFloat32 || !t->is_contiguous() || t->numel() != e.param.numel() ||
			    t->device() != e.param.device())
				throw std::runtime_error("adamStepMulti needs contiguous float32 tensors of one size and device per entry");
		gsr_adam_multi_tensor m{};
		m.param = e.param.data_ptr<float>(); m.grad = e.grad.data_ptr<float>();
		m.exp_avg = e.exp_avg.data_ptr<float>(); m.exp_avg_sq = e.exp_avg_sq.data_ptr<float>();
		m.n = e.param.numel(); m.lr = e.lr; m.step = e.step; m.grad_scale = e.grad_scale;
		ts.push_back(m);
	}
	check(gsr_adam_step_multi(static_cast<int>(ts.size()), ts.data(), beta1, beta2, eps, current_stream(entries[0].param)),
	      "adamStepMulti");
}

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix)
{
	const int P = static_cast<int>(means3D.size(0));
	torch::Tensor present = torch::zeros({P}, means3D.options().dtype(at::kBool));
	if (P != 0) {
		F32 m3(means3D), view(viewmatrix), proj(projmatrix);
		check(gsr_mark_visible(P, m3.ptr, view.ptr, proj.ptr, reinterpret_cast<uint8_t*>(present.data_ptr<bool>()),
		                       current_stream(means3D)),
		      "markVisible");
	}
	return present;
}
