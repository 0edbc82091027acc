/*
 * oracle/ref_host/sophus_standin.h -- STAND-INS (ours) for the two third-party types GaussianModel::applyScaledTransformation's
 * SIGNATURE needs (src/gaussian_model.cpp:379-396: `const Sophus::SE3f T`, `tensor_utils::EigenMatrix2TorchTensor(T.matrix(),
 * device_type_)`), TEST INFRASTRUCTURE ONLY.  Sophus, Eigen and the reference's include/tensor_utils.h (OpenCV) do not exist in
 * this image.  The function BODY is the reference's, extracted verbatim; what is restated here is only "a rigid transform that
 * hands out its 4x4 matrix" and "that matrix as a row-major float tensor on a device" (include/tensor_utils.h:180-194).
 */
#pragma once
#include <torch/torch.h>

namespace Eigen {
struct StandInMatrix4f {
	float m[4][4];   /* m[r][c] */
};
}  // namespace Eigen

namespace Sophus {
class SE3f {
public:
	SE3f()
	{
		for (int r = 0; r < 4; r++)
			for (int c = 0; c < 4; c++) M_.m[r][c] = r == c ? 1.f : 0.f;
	}
	explicit SE3f(const Eigen::StandInMatrix4f& M) : M_(M) {}
	Eigen::StandInMatrix4f matrix() const { return M_; }

private:
	Eigen::StandInMatrix4f M_;
};
}  // namespace Sophus

namespace tensor_utils {
/* include/tensor_utils.h:180-194: the matrix as a [rows, cols] float tensor (element (r, c) at [r][c]) on the device */
inline torch::Tensor EigenMatrix2TorchTensor(Eigen::StandInMatrix4f eigen_matrix, torch::DeviceType device_type = torch::kCUDA)
{
	torch::Tensor tensor = torch::from_blob(&eigen_matrix.m[0][0], {4, 4}, torch::TensorOptions().dtype(torch::kFloat)).clone();
	return tensor.to(device_type);
}
}  // namespace tensor_utils
