// ply_io.cpp -- GaussianModel::savePly / loadPly of the LibTorch host (src/gaussian_model.cpp:838-1047 of the reference).
//
// The reference writes through tinyply (third_party/tinyply): one `vertex` element of float32 properties in the order
//   x y z  nx ny nz  f_dc_0..2  f_rest_0..(3 (M-1) - 1)  opacity  scale_0..2  rot_0..3
// binary little endian, the RAW (pre-activation) parameters, f_dc / f_rest channel-major (features.transpose(1,2).flatten(1)),
// normals zero.  This writer produces the same bytes (tests/test_points_and_ply.py compares against a file written by the
// reference's own savePly) without tinyply; the reader takes any property order and ignores properties it does not know, as
// tinyply's request_properties_from_element does, and like the reference sets the active SH degree to the maximum (:1046).
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "gaussian_model_lite.h"

namespace {
std::vector<std::string> property_names(int64_t n_rest)
{
	std::vector<std::string> n = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"};
	for (int64_t i = 0; i < n_rest; i++) n.push_back("f_rest_" + std::to_string(i));
	n.push_back("opacity");
	for (int i = 0; i < 3; i++) n.push_back("scale_" + std::to_string(i));
	for (int i = 0; i < 4; i++) n.push_back("rot_" + std::to_string(i));
	return n;
}
}  // namespace

void GaussianModel::savePly(const std::string& result_path)
{
	torch::NoGradGuard ng;
	syncFeatures();
	const int64_t P = xyz_.size(0), M = features_.size(1);
	auto cpu = [](const torch::Tensor& t) { return t.detach().to(torch::kCPU, torch::kFloat32).contiguous(); };
	auto xyz = cpu(xyz_);
	auto feat = cpu(features_);                                                          // [P, M, 3], dc first
	auto f_dc = feat.slice(1, 0, 1).transpose(1, 2).reshape({P, 3});
	auto f_rest = feat.slice(1, 1, M).transpose(1, 2).reshape({P, 3 * (M - 1)});
	auto rows = torch::cat({xyz, torch::zeros_like(xyz), f_dc, f_rest, cpu(opacity_).reshape({P, 1}), cpu(scaling_), cpu(rotation_)}, 1)
	                .contiguous();
	std::ofstream os(result_path, std::ios::out | std::ios::binary);
	if (!os) throw std::runtime_error("failed to open " + result_path);
	os << "ply\nformat binary_little_endian 1.0\nelement vertex " << P << "\n";
	for (const auto& n : property_names(3 * (M - 1))) os << "property float " << n << "\n";
	os << "end_header\n";
	os.write(reinterpret_cast<const char*>(rows.data_ptr<float>()), static_cast<std::streamsize>(rows.numel() * sizeof(float)));
	if (!os) throw std::runtime_error("failed to write " + result_path);
}

void GaussianModel::loadPly(const std::string& ply_path)
{
	torch::NoGradGuard ng;
	std::ifstream is(ply_path, std::ios::binary);
	if (!is.is_open() || is.fail()) throw std::runtime_error("Fail to open ply file at " + ply_path);   // the reference's text, :843
	std::string line;
	std::getline(is, line);
	if (line.rfind("ply", 0) != 0) throw std::runtime_error("not a PLY file: " + ply_path);
	int64_t count = -1;
	bool in_vertex = false, binary_le = false;
	std::vector<std::string> props;
	while (std::getline(is, line)) {
		std::istringstream ls(line);
		std::string tok;
		ls >> tok;
		if (tok == "end_header") break;
		if (tok == "format") {
			ls >> tok;
			binary_le = tok == "binary_little_endian";
		} else if (tok == "element") {
			std::string name;
			int64_t n;
			ls >> name >> n;
			in_vertex = name == "vertex";
			if (in_vertex) count = n;
			else if (count < 0) throw std::runtime_error("PLY: elements before `vertex` are not supported");
		} else if (tok == "property" && in_vertex) {
			std::string type, name;
			ls >> type >> name;
			if (type != "float" && type != "float32") throw std::runtime_error("PLY: unexpected property type " + type);
			props.push_back(name);
		}
	}
	if (!binary_le) throw std::runtime_error("PLY: only binary_little_endian is supported (what savePly writes)");
	if (count < 0) throw std::runtime_error("PLY: no vertex element");
	const int64_t C = static_cast<int64_t>(props.size());
	auto data = torch::empty({count, C}, torch::kFloat32);
	is.read(reinterpret_cast<char*>(data.data_ptr<float>()), static_cast<std::streamsize>(count * C * sizeof(float)));
	if (is.gcount() != static_cast<std::streamsize>(count * C * sizeof(float))) throw std::runtime_error("PLY: truncated payload");
	std::unordered_map<std::string, int64_t> col;
	for (int64_t i = 0; i < C; i++) col[props[i]] = i;
	auto take = [&](const std::vector<std::string>& names) {
		std::vector<int64_t> idx;
		for (const auto& n : names) {
			auto it = col.find(n);
			if (it == col.end()) throw std::runtime_error("PLY: missing property " + n);
			idx.push_back(it->second);
		}
		return data.index_select(1, torch::tensor(idx, torch::kLong));
	};
	const int64_t M = static_cast<int64_t>(max_sh_degree_ + 1) * (max_sh_degree_ + 1), n_rest = 3 * (M - 1);
	std::vector<std::string> rest_names, scale_names, rot_names;
	for (int64_t i = 0; i < n_rest; i++) rest_names.push_back("f_rest_" + std::to_string(i));
	for (int i = 0; i < 3; i++) scale_names.push_back("scale_" + std::to_string(i));
	for (int i = 0; i < 4; i++) rot_names.push_back("rot_" + std::to_string(i));
	auto f_dc = take({"f_dc_0", "f_dc_1", "f_dc_2"}).reshape({count, 3, 1}).transpose(1, 2);
	auto f_rest = take(rest_names).reshape({count, 3, M - 1}).transpose(1, 2);
	const auto dev = xyz_.defined() ? xyz_.device() : device_;
	auto leaf = [&](torch::Tensor t) { return t.contiguous().to(dev).set_requires_grad(true); };
	xyz_ = leaf(take({"x", "y", "z"}));
	features_row_step_ = torch::Tensor();   // a new SH tensor: no lazy state
	features_lr_hist_.clear();
	features_ = leaf(torch::cat({f_dc, f_rest}, 1));   // one [P, M, 3] leaf (the reference: features_dc_ | features_rest_)
	opacity_ = leaf(take({"opacity"}));
	scaling_ = leaf(take(scale_names));
	rotation_ = leaf(take(rot_names));
	const auto o = xyz_.options().requires_grad(false);
	max_radii2D_ = torch::zeros({count}, o);
	xyz_gradient_accum_ = torch::zeros({count, 1}, o);
	denom_ = torch::zeros({count, 1}, o);
	groups_.clear();
	active_sh_degree_ = max_sh_degree_;   // :1046
}
