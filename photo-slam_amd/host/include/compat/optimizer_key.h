// compat/optimizer_key.h -- the one LibTorch API drift "unchanged" Photo-SLAM host code meets on a current LibTorch
// (last row).
//
// src/gaussian_model.cpp:571, 598, 670 (replaceTensorToOptimizer, prunePoints, densificationPostfix) key the Adam state map
// with   c10::guts::to_string(param.unsafeGetTensorImpl())   -- valid for LibTorch <= 2.1.2 (README.md:82), whose
// OptimizerParamState map is keyed by std::string.  From LibTorch 2.2 on the map is
//   ska::flat_hash_map<void*, std::unique_ptr<OptimizerParamState>>      (torch/optim/optimizer.h)
// and c10::guts::to_string no longer exists.  The three call sites (six occurrences) become
//
//   auto key = optim_key(param);                  // instead of c10::guts::to_string(param.unsafeGetTensorImpl())
//
// and compile against either generation.  Nothing on the rasterizer side of the boundary changes.
//
// ZERO-EDIT FORM.  All six occurrences have the shape  `auto key = c10::guts::to_string(impl); ... state[key] / find(key) /
// erase(key); key = c10::guts::to_string(impl2);`  -- the key's type is never spelled.  On a LibTorch whose state map is keyed
// by pointer this header therefore also supplies the missing  c10::guts::to_string(c10::TensorImpl*)  as an overload that
// returns the pointer itself (next to `using std::to_string`, c10/util/string_utils.h, which has no pointer overload): with
//     target_compile_options(gaussian_mapper PRIVATE -include compat/optimizer_key.h)
// src/gaussian_model.cpp compiles UNCHANGED.  oracle/build_ref.py compiles the reference's own member functions exactly so
// (no rewrite), as the checker of this repository's densification and as the model of the drop-in train step
// (tests/test_reference_host.py).
#pragma once
#include <torch/torch.h>

#include <string>
#include <type_traits>

namespace photoslam_compat {
// the key type of torch::optim::Optimizer::state() of the LibTorch being compiled against
using optim_state_map = std::remove_reference_t<decltype(std::declval<torch::optim::Optimizer&>().state())>;
using optim_key_t = typename optim_state_map::key_type;

template <typename Key = optim_key_t>
inline Key optim_key(const at::Tensor& param)
{
	if constexpr (std::is_same_v<Key, std::string>)
		return std::to_string(reinterpret_cast<uintptr_t>(param.unsafeGetTensorImpl()));   // what c10::guts::to_string produced a key from
	else
		return static_cast<Key>(param.unsafeGetTensorImpl());
}
}  // namespace photoslam_compat

using photoslam_compat::optim_key;

namespace c10 { namespace guts {
// only where the state map is NOT keyed by std::string (LibTorch >= 2.2); older LibTorch has its own c10::guts::to_string
template <typename Key = photoslam_compat::optim_key_t, std::enable_if_t<!std::is_same_v<Key, std::string>, int> = 0>
inline Key to_string(c10::TensorImpl* impl)
{
	return static_cast<Key>(impl);
}
}}  // namespace c10::guts
