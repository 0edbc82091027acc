// spatial.cpp -- distCUDA2 (replaces third_party/simple-knn/spatial.cu:15-26 of the reference) on top of
// gsr_knn_mean_dist2 (include/gsr.h, csrc/knn.hip).  A translation unit of its own because the reference ships it as a
// library of its own (`simple_knn`, CMakeLists.txt:54-60) that gaussian_mapper links by name (:145-152).
#include "spatial.h"

#include <stdexcept>
#include <string>

#include "../../../include/gsr.h"

#ifndef GSR_HOST_NO_HIP
#include <c10/hip/HIPStream.h>
#endif

namespace {
// the kNN's scratch (Morton codes, sort ping/pong, box tables) lives in a caller-owned byte tensor, grown on demand
char* resize_tensor(void* ctx, size_t bytes)
{
	auto* t = static_cast<torch::Tensor*>(ctx);
	t->resize_({static_cast<int64_t>(bytes)});
	return reinterpret_cast<char*>(t->data_ptr());
}
void* stream_of(const torch::Tensor& t)
{
#ifndef GSR_HOST_NO_HIP
	if (t.is_cuda()) return c10::hip::getCurrentHIPStream(t.device().index()).stream();
#endif
	return nullptr;
}
}  // namespace

torch::Tensor distCUDA2(const torch::Tensor& points)
{
	const int P = static_cast<int>(points.size(0));
	torch::Tensor means = torch::zeros({P}, points.options().dtype(torch::kFloat32));
	if (P != 0) {
		torch::Tensor pts = points.contiguous();
		torch::Tensor scratch = torch::empty({0}, points.options().dtype(torch::kByte));
		const int st = gsr_knn_mean_dist2(P, pts.data_ptr<float>(), means.data_ptr<float>(), resize_tensor, &scratch, stream_of(points));
		if (st != GSR_OK) {
			std::string msg = std::string("distCUDA2: ") + gsr_strerror(st);
			if (st == GSR_ERR_HIP) msg += std::string(" [") + gsr_last_hip_error_string() + "]";
			throw std::runtime_error(msg);
		}
	}
	return means;
}
