// tests/compat: a consumer of host/include/compat/optimizer_key.h (tests/test_compat_optimizer_key.py)
#include "compat/optimizer_key.h"
int main() {
	auto p = torch::zeros({3}).requires_grad_();
	torch::optim::Adam opt({p}, torch::optim::AdamOptions(0.1));
	p.mutable_grad() = torch::ones({3});
	opt.step();
	auto& state = opt.state();
	auto key = optim_key(opt.param_groups()[0].params()[0]);
	return state.find(key) != state.end() ? 0 : 1;
}
