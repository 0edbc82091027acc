// keyframe_batch_exchange.cpp -- see keyframe_batch_exchange.h; the data-parallel half of TrainStep lives here too.
#include "keyframe_batch_exchange.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <stdexcept>

#include "gaussian_model_lite.h"

#ifndef GSR_HOST_NO_HIP
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#endif

torch::Tensor oneBuffer(const std::vector<torch::Tensor>& tensors)
{
	if (tensors.size() < 2) return torch::Tensor();
	for (const auto& t : tensors)
		if (!t.defined() || !t.is_contiguous() || t.scalar_type() != torch::kFloat32) return torch::Tensor();
	const auto& storage = tensors[0].storage();
	for (const auto& t : tensors)
		if (t.storage().data() != storage.data()) return torch::Tensor();
	std::vector<std::pair<int64_t, int64_t>> spans;
	for (const auto& t : tensors) spans.emplace_back(t.storage_offset(), t.numel());
	std::sort(spans.begin(), spans.end());
	int64_t end = spans[0].first;
	for (const auto& s : spans) {
		if (s.first != end) return torch::Tensor();
		end = s.first + s.second;
	}
	if (spans[0].first != 0 || static_cast<size_t>(end) * 4 != storage.nbytes()) return torch::Tensor();
	auto flat = torch::empty({0}, tensors[0].options().requires_grad(false));
	flat.set_(storage, 0, {end}, {1});
	return flat;
}

#ifndef GSR_HOST_NO_HIP
// The current stream of this thread switched to `side` for a scope -- restored on every way out, an exception included (a
// collective or a pack launch that throws must not leave the caller's later torch ops on the gather stream).
struct CurrentStreamScope {
	c10::hip::HIPStreamMasqueradingAsCUDA prev;
	explicit CurrentStreamScope(const c10::hip::HIPStreamMasqueradingAsCUDA& side)
	    : prev(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(side.device_index()))
	{
		c10::hip::setCurrentHIPStreamMasqueradingAsCUDA(side);
	}
	~CurrentStreamScope() { c10::hip::setCurrentHIPStreamMasqueradingAsCUDA(prev); }
	CurrentStreamScope(const CurrentStreamScope&) = delete;
	CurrentStreamScope& operator=(const CurrentStreamScope&) = delete;
};
#endif

// Device tensors over gloo travel through the host and block the calling thread in Work::wait(): refused, except for the
// functional check that runs several ranks on ONE GPU (tools/gpu.sh share2 sets GSR_EXCHANGE_ALLOW_GLOO_DEVICE=1).
static bool glooDeviceTensorsAllowed()
{
	const char* e = std::getenv("GSR_EXCHANGE_ALLOW_GLOO_DEVICE");
	return e && e[0] == '1';
}

GradientReduction::GradientReduction(c10::intrusive_ptr<c10d::ProcessGroup> pg, std::vector<torch::Tensor> tensors, bool sum_only)
    : pg_(std::move(pg)), tensors_(std::move(tensors)), sum_only_(sum_only)
{
	avg_ = !sum_only_ && pg_->getBackendName() == "nccl";
	const int n = static_cast<int>(tensors_.size());
	order_.resize(n);
	for (int i = 0; i < n; i++) order_[i] = i;
	std::stable_sort(order_.begin(), order_.end(), [&](int a, int b) { return tensors_[a].numel() > tensors_[b].numel(); });
	group_of_.assign(n, -1);
	// members of one buffer share a single collective
	std::map<const void*, std::vector<int>> by_storage;
	std::vector<const void*> keys;
	for (int i : order_) {
		const void* k = tensors_[i].storage().data();
		if (!by_storage.count(k)) keys.push_back(k);
		by_storage[k].push_back(i);
	}
	for (const void* k : keys) {
		const auto& members = by_storage[k];
		std::vector<torch::Tensor> ts;
		for (int i : members) ts.push_back(tensors_[i]);
		auto flat = oneBuffer(ts);
		if (flat.defined()) {
			for (int i : members) group_of_[i] = static_cast<int>(groups_.size());
			groups_.push_back({flat, nullptr, false});
		}
	}
	for (int i : order_)
		if (group_of_[i] < 0) {
			group_of_[i] = static_cast<int>(groups_.size());
			groups_.push_back({tensors_[i], nullptr, false});
		}
	c10d::AllreduceOptions opts;
	opts.reduceOp = avg_ ? c10d::ReduceOp::AVG : c10d::ReduceOp::SUM;
	for (int i : order_) {   // issue in size order; a shared buffer goes out when its first member comes up
		auto& grp = groups_[static_cast<size_t>(group_of_[i])];
		if (grp.work) continue;
		std::vector<at::Tensor> v{grp.flat};
		grp.work = pg_->allreduce(v, opts);
	}
}

void GradientReduction::wait(int i)
{
	auto& grp = groups_.at(static_cast<size_t>(group_of_.at(static_cast<size_t>(i))));
	grp.work->wait();
	if (!avg_ && !sum_only_ && !grp.scaled) {
		grp.flat.mul_(1.0 / pg_->getSize());
		grp.scaled = true;
	}
}

void GradientReduction::waitAll()
{
	for (int i : order_) wait(i);
}

ViewFactoredExchange::ViewFactoredExchange(c10::intrusive_ptr<c10d::ProcessGroup> pg, torch::Tensor send, torch::Tensor camera_center,
                                           std::vector<torch::Tensor> others, void* gather_stream, torch::Tensor gathered)
    : pg_(std::move(pg))
{
	const int64_t N = pg_->getSize(), P = send.size(0) - 1;
	const auto o = send.options().requires_grad(false);
	if (pg_->getBackendName() == "gloo" && send.is_cuda() && !glooDeviceTensorsAllowed())
		throw std::runtime_error("ViewFactoredExchange: gloo moves host tensors; use the RCCL backend for device tensors");
	// ONE all-gather: rows 0 .. P-1 of `send` are this view's colour gradients, row P its camera centre -- every collective
	// costs a launch on RCCL's stream and two cross-stream hand-offs (measured at one rank: four collectives per step cost
	// ~0.1 ms more than two), which is more than the overlap of a second row range's rebuild with its gather could win back
	// (row P: the camera centre.  TrainStep::renderAndBackward wrote it BEFORE backward -- a gather that does not wait for the
	// end of the backward pass must not depend on a copy queued behind it; other callers get it written here)
	if (!gather_stream) send.select(0, P).copy_(camera_center.detach().reshape({3}).to(o));
	const bool own_buffer = !(gathered.defined() && gathered.size(0) == N && gathered.size(1) == P + 1 && gathered.is_contiguous() &&
	                          gathered.device() == send.device());
	gathered_ = own_buffer ? torch::empty({N, P + 1, 3}, o) : gathered;
	auto in = send.unsqueeze(0);   // ([1, P + 1, 3]: gloo checks the input against a 1/N chunk of the output)
	Part p;
	p.row0 = 0;
	p.views = gathered_.narrow(1, 0, P);      // [N, P, 3], view stride (P + 1) * 3
#ifndef GSR_HOST_NO_HIP
	if (gather_stream && send.is_cuda()) {
		// issue the gather with the side stream current: ProcessGroupNCCL orders the collective behind the CURRENT stream, and
		// this one waits only for "dL_dcolor_view is complete" (gsr_backward), not for the kernels queued behind that point.
		// Both buffers were allocated on the compute stream: the caching allocator is told that the side stream uses them.
		const auto idx = send.device().index();
		auto side = c10::hip::getStreamFromExternalMasqueradingAsCUDA(static_cast<hipStream_t>(gather_stream), idx);
		send.record_stream(side.unwrap());
		gathered_.record_stream(side.unwrap());
		const auto prev = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(idx);
		if (own_buffer) {
			// allocated just now, behind the enqueued backward pass: the block may be one the pass has released and still writes --
			// the second stream waits for the compute stream's tail before RCCL touches it (the caller-provided buffer of
			// TrainStep needs no such wait: it existed before the pass was enqueued)
			at::cuda::CUDAEvent tail;
			tail.record(prev);
			tail.block(side);
		}
		CurrentStreamScope on_side(side);
		p.work = pg_->_allgather_base(gathered_, in);
	} else
#endif
		p.work = pg_->_allgather_base(gathered_, in);
	parts_.push_back(p);
	centres_ = gathered_.select(1, P);       // [N, 3], stride (P + 1) * 3
	// (summed, not averaged: finishGeomAdam() multiplies by 1/N as it reads the gradients)
	reduction_ = std::make_unique<GradientReduction>(pg_, std::move(others), /*sum_only=*/true);
}

ViewFactoredExchange::ViewFactoredExchange(c10::intrusive_ptr<c10d::ProcessGroup> pg, Packed pk, torch::Tensor camera_center,
                                           std::vector<torch::Tensor> others, void* gather_stream)
    : pg_(std::move(pg))
{
	const int64_t N = pg_->getSize(), P = pk.color_view.size(0);
	if (pg_->getBackendName() == "gloo" && pk.color_view.is_cuda() && !glooDeviceTensorsAllowed())
		throw std::runtime_error("ViewFactoredExchange: gloo moves host tensors; use the RCCL backend for device tensors");
	const int64_t words = packedViewWords(P, pk.capacity);
	if (pk.send.numel() < words || pk.gathered.numel() < N * words)
		throw std::runtime_error("ViewFactoredExchange: the packed buffers are too small for this capacity");
	auto send = pk.send.narrow(0, 0, words);
	gathered_ = pk.gathered.narrow(0, 0, N * words);
	Part p;
	p.msg_stride = words;
	p.capacity = pk.capacity;
	p.messages = gathered_;
	auto issue = [&]() {
		// (four launches on the current stream -- or none: the backward pass wrote the message itself, ShAdamStep::packed_view)
		if (!pk.prepacked) packColorView(pk.color_view, camera_center.detach().reshape({3}), pk.capacity, send, pk.scratch);
		p.work = pg_->_allgather_base(gathered_, send);
	};
#ifndef GSR_HOST_NO_HIP
	if (gather_stream && pk.color_view.is_cuda()) {
		// (as the dense form: the stream that waits only for "dL_dcolor_view is complete" is made current -- the message is
		// built there and ProcessGroupNCCL orders the gather behind it)
		const auto idx = pk.color_view.device().index();
		auto side = c10::hip::getStreamFromExternalMasqueradingAsCUDA(static_cast<hipStream_t>(gather_stream), idx);
		// (allocated on the compute stream, read and written on this one: the caching allocator is told, as for the dense buffers)
		pk.color_view.record_stream(side.unwrap());
		pk.send.record_stream(side.unwrap());
		pk.gathered.record_stream(side.unwrap());
		CurrentStreamScope on_side(side);
		issue();
	} else
#endif
		issue();
	parts_.push_back(p);
	reduction_ = std::make_unique<GradientReduction>(pg_, std::move(others), /*sum_only=*/true);
}

torch::Tensor ViewFactoredExchange::centres()
{
	part(0);   // (the centres travel with the colour gradients)
	return centres_;
}

const ViewFactoredExchange::Part& ViewFactoredExchange::part(int k)
{
	auto& p = parts_.at(static_cast<size_t>(k));
	if (p.work) {
		p.work->wait();
		p.work = nullptr;
	}
	return p;
}

void ViewFactoredExchange::waitAll()
{
	for (int k = 0; k < parts(); k++) part(k);
	reduction_->waitAll();
}

// ---- TrainStep: the data-parallel step --------------------------------------------------------------------------------------

TrainStep::~TrainStep()
{
#ifndef GSR_HOST_NO_HIP
	if (counts_event_) (void)hipEventDestroy(static_cast<hipEvent_t>(counts_event_));
	for (auto& e : wait_events_)
		if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
#endif
}

// profile_exchange_: HIP events on the compute stream in front of and behind a Work::wait() -- the elapsed time between the two
// is the time the stream idled for the collective (0 when it had landed already): the EXPOSED communication of the step.
void TrainStep::markWait(int k)
{
#ifndef GSR_HOST_NO_HIP
	if (!profile_exchange_ || !gaussians_->xyz_.is_cuda()) return;
	auto& ev = wait_events_[k];
	if (!ev) {
		hipEvent_t e = nullptr;
		if (hipEventCreate(&e) != hipSuccess) return;
		ev = e;
	}
	(void)hipEventRecord(static_cast<hipEvent_t>(ev), c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(gaussians_->xyz_.device().index()).stream());
#else
	(void)k;
#endif
}

std::vector<double> TrainStep::exchangeWaitMs()
{
	// {gather wait, reduce wait, start -> gather wait (forward, loss, backward), SH step between the two waits, reduce wait -> end}
	std::vector<double> out{-1.0, -1.0, -1.0, -1.0, -1.0};
#ifndef GSR_HOST_NO_HIP
	const int pairs[5][2] = {{0, 1}, {2, 3}, {4, 0}, {1, 2}, {3, 5}};
	for (int j = 0; j < 5; j++) {
		auto a = static_cast<hipEvent_t>(wait_events_[pairs[j][0]]), b = static_cast<hipEvent_t>(wait_events_[pairs[j][1]]);
		float ms = 0.f;
		if (a && b && hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) out[static_cast<size_t>(j)] = ms;
	}
#endif
	return out;
}

// The packed exchange: every rank's message must have room for the LARGEST view of the batch.  A view's row count is its
// number of visible Gaussians, which the forward pass leaves on the HOST (gsr_last_visible_count) -- so the ranks agree on the
// maximum host-side: one int per rank over a host process group (setCountGroup: gloo), issued right behind the forward pass
// and picked up after the host has queued the loss and the backward pass.  The host runs ~0.6 ms ahead of the device there
// (bench.py: host.blocked_in_forward_sync_us_per_step), a gloo all-gather of N ints on one node takes a fraction of that, and
// the device is not involved at all.  (Round 4 first sent the counts through RCCL on the gather stream -- pinned copy in, gather,
// pinned copy out: measured +42 us per step ON THE COMPUTE STREAM at one rank, profiles/r04_r: the two host<->device copies cost the
// kernels running next to them more than the 12 MB they save on the links.  That route remains for a caller without a host group.)
// One rank needs no exchange.
void TrainStep::beginCountExchange()
{
	torch::NoGradGuard ng;
	const int64_t N = process_group_->getSize();
	const int V = lastVisibleCount();
	// (-1: no forward pass has run on this thread -- a capacity agreed on from it would drop every row of this view's message)
	if (V < 0) throw std::runtime_error("beginCountExchange: no forward pass on this thread has left a visible count (gsr_last_visible_count() < 0)");
	// a caller that drives renderAndBackward() / finishOneIteration() itself never reaches finishCountExchange(): the previous
	// step's exchange is finished here before the next collective is issued on the same group (and before the pinned word of the
	// device route is rewritten under its copy)
	if (count_work_ || counts_on_device_route_) (void)finishCountExchange();
	count_work_ = nullptr;
	counts_on_device_route_ = false;
	if (N == 1) {
		counts_host_ = torch::full({1}, V, torch::kInt32);
		return;
	}
	const auto& xyz = gaussians_->xyz_;
	if (count_group_ || !xyz.is_cuda()) {   // host tensors over a host group (gloo); in flight until finishCountExchange()
		auto& pg = count_group_ ? count_group_ : process_group_;
		count_own_host_ = torch::full({1}, V, torch::kInt32);
		counts_host_ = torch::empty({N}, torch::kInt32);
		count_work_ = pg->_allgather_base(counts_host_, count_own_host_);
		return;
	}
#ifndef GSR_HOST_NO_HIP
	counts_on_device_route_ = true;
	const auto idx = xyz.device().index();
	if (!gather_stream_) gather_stream_ = c10::hip::getStreamFromPool(/*isHighPriority=*/false, idx).stream();
	if (!count_own_pinned_.defined() || !counts_host_.defined() || counts_host_.numel() != N || !counts_host_.is_pinned()) {
		count_own_pinned_ = torch::empty({1}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
		counts_host_ = torch::empty({N}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
		count_own_dev_ = torch::empty({1}, xyz.options().dtype(torch::kInt32).requires_grad(false));
		counts_dev_ = torch::empty({N}, xyz.options().dtype(torch::kInt32).requires_grad(false));
	}
	if (!counts_event_) {
		hipEvent_t e = nullptr;
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) throw std::runtime_error("beginCountExchange: hipEventCreate failed");
		counts_event_ = e;
	}
	count_own_pinned_.data_ptr<int32_t>()[0] = V;   // (the previous step's copy was waited for in finishCountExchange)
	auto side = c10::hip::getStreamFromExternalMasqueradingAsCUDA(static_cast<hipStream_t>(gather_stream_), idx);
	CurrentStreamScope on_side(side);
	count_own_dev_.copy_(count_own_pinned_, /*non_blocking=*/true);
	process_group_->_allgather_base(counts_dev_, count_own_dev_)->wait();   // (stream-side: the gather stream waits, not the host)
	counts_host_.copy_(counts_dev_, /*non_blocking=*/true);
	(void)hipEventRecord(static_cast<hipEvent_t>(counts_event_), side.stream());
#endif
}

// The mask and prefix sections of this view's message from the radii the forward pass has just left (gsr_forward returns
// behind preprocess_fwd: they are complete) -- three small launches on the gather stream, next to the forward blend; the compute
// stream waits for them in front of the backward pass, which writes rows and header (ShAdamStep::packed_view).
void TrainStep::planPackedView(const torch::Tensor& radii)
{
	torch::NoGradGuard ng;
	const int64_t cap = (radii.size(0) + 3) / 4 * 4;
#ifndef GSR_HOST_NO_HIP
	if (radii.is_cuda()) {
		const auto idx = radii.device().index();
		if (!gather_stream_) gather_stream_ = c10::hip::getStreamFromPool(/*isHighPriority=*/false, idx).stream();
		auto side = c10::hip::getStreamFromExternalMasqueradingAsCUDA(static_cast<hipStream_t>(gather_stream_), idx);
		const auto prev = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(idx);
		radii.record_stream(side.unwrap());
		sh_packed_send_.record_stream(side.unwrap());
		if (!sh_pack_scratch_.defined()) {   // (allocated on the compute stream, used on the gather stream from here on)
			packViewPlan(radii, cap, sh_packed_send_, sh_pack_scratch_);   // first step: on the compute stream, allocates the scratch
			sh_pack_scratch_.record_stream(side.unwrap());
			return;
		}
		at::cuda::CUDAEvent planned;
		{
			CurrentStreamScope on_side(side);
			packViewPlan(radii, cap, sh_packed_send_, sh_pack_scratch_);
			planned.record(side);
		}
		planned.block(prev);
		return;
	}
#endif
	packViewPlan(radii, cap, sh_packed_send_, sh_pack_scratch_);
}

int64_t TrainStep::finishCountExchange()
{
	if (count_work_) {   // the host group's all-gather: issued behind the forward pass, long landed
		count_work_->wait();
		count_work_ = nullptr;
	}
#ifndef GSR_HOST_NO_HIP
	if (counts_on_device_route_ && counts_event_ && hipEventSynchronize(static_cast<hipEvent_t>(counts_event_)) != hipSuccess)
		throw std::runtime_error("finishCountExchange: waiting for the visible counts failed");
#endif
	counts_on_device_route_ = false;   // (landed: nothing in flight any more)
	int64_t most = 0;
	const int32_t* c = counts_host_.data_ptr<int32_t>();
	for (int64_t i = 0; i < counts_host_.numel(); i++) most = std::max<int64_t>(most, c[i]);
	return (most + 3) / 4 * 4;   // (a multiple of 4 rows: the messages stay 16-byte aligned)
}

void TrainStep::setProcessGroup(c10::intrusive_ptr<c10d::ProcessGroup> pg, bool factored)
{
	process_group_ = std::move(pg);
	factored_exchange_ = process_group_ && factored;
	if (process_group_) fused_sh_adam_ = false;   // the optimizer follows the gradient exchange
}

torch::Tensor TrainStep::trainForOneIterationDataParallel(std::shared_ptr<GaussianKeyframe> kf, torch::Tensor gt_image, torch::Tensor mask)
{
	if (!process_group_) throw std::runtime_error("trainForOneIterationDataParallel: setProcessGroup() first");
	markWait(4);
	auto loss = renderAndBackward(kf, gt_image, mask);
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	const auto& o = g->opt_;
	auto params = g->paramsRaw();
	// keyframe-batch data parallelism: mean of the per-view gradients over the ranks, in flight from here on
	std::unique_ptr<ViewFactoredExchange> vf;
	std::unique_ptr<GradientReduction> red;
	if (factored_exchange_) {
		std::vector<torch::Tensor> others;
		for (int i : {0, 2, 3, 4}) others.push_back(params[static_cast<size_t>(i)].grad());
		if (packed_this_step_) {
			ViewFactoredExchange::Packed pk;
			pk.color_view = sh_grad_view_;
			pk.send = sh_packed_send_;
			pk.gathered = sh_packed_gathered_;
			pk.scratch = sh_pack_scratch_;
			pk.capacity = finishCountExchange();
			pk.prepacked = prepacked_this_step_;
			vf = std::make_unique<ViewFactoredExchange>(process_group_, pk, kf->camera_center_, others,
			                                            gather_stream_in_use_ ? gather_stream_ : nullptr);
			sh_pack_scratch_ = pk.scratch;   // (grown on first use: kept)
		} else
			vf = std::make_unique<ViewFactoredExchange>(process_group_, sh_send_, kf->camera_center_, others,
			                                            gather_stream_in_use_ ? gather_stream_ : nullptr, sh_gathered_);
	} else {
		std::vector<torch::Tensor> grads;
		for (auto& p : params) grads.push_back(p.grad());
		red = std::make_unique<GradientReduction>(process_group_, grads);
	}
	auto wait_all = [&]() {
		if (vf) vf->waitAll();
		else red->waitAll();
	};
	if (densifyDue()) {
		// every tensor is about to be rebuilt and this step's update is skipped: the exchange only has to finish.  The
		// statistics accumulated PER RANK since the last densification (SUM and MAX commute with the accumulation over
		// iterations): norm sums and counts SUM, radii MAX -- every rank then takes the same decisions with the same samples.
		wait_all();
		c10d::AllreduceOptions sum, mx;
		sum.reduceOp = c10d::ReduceOp::SUM;
		mx.reduceOp = c10d::ReduceOp::MAX;
		std::vector<at::Tensor> a{g->xyz_gradient_accum_}, d{g->denom_}, r{g->max_radii2D_};
		auto w1 = process_group_->allreduce(a, sum), w2 = process_group_->allreduce(d, sum), w3 = process_group_->allreduce(r, mx);
		w1->wait();
		w2->wait();
		w3->wait();
		finishBegin();   // densifies (and drops the gradients: the fresh leaves have none)
		finishEnd();
		return loss;
	}
	finishBegin();   // (an opacity reset alone replaces ONE leaf: the others keep their gradients and step below)
	if (iteration_ < o.iterations_) {
		if (vf) {
			if (g->features_.size(1) == 16 && g->groups_.size() > 1) {
				// the SH gradient is rebuilt from the gathered views (reads xyz_: before ITS update) and applied part by part as
				// each all-gather lands, while the all-reduce of the other four tensors is on the links
				for (int k = 0; k < vf->parts(); k++) {
					markWait(0);   // (profile_exchange_: how long does the compute stream wait for the all-gather?)
					const auto& p = vf->part(k);
					markWait(1);
					if (p.msg_stride) {
						// the first packed steps of this object (and every one under GSR_CHECK_PACKED=1) read the gathered headers back:
						// a rank whose message was written for another P / capacity, or that dropped rows, stops the run here
						static const int check_env = [] { const char* e = getenv("GSR_CHECK_PACKED"); return (e && *e) ? atoi(e) : -1; }();
						if (check_env > 0 || (check_env < 0 && packed_steps_checked_ < check_packed_first_steps_)) {
							checkPackedViews(p.messages, p.msg_stride, process_group_->getSize(), g->xyz_.size(0), p.capacity);
							packed_steps_checked_++;
						}
						stepFeaturesFromPackedViews(p.messages, p.msg_stride, process_group_->getSize());
					}
					else stepFeaturesFromViews(vf->centres(), p.views, p.row0, k == 0);
				}
				finishFeaturesFromViews();
			} else {   // other SH layouts: gradient tensor + separate pass (whole batch)
				std::vector<torch::Tensor> views;
				for (int k = 0; k < vf->parts(); k++) views.push_back(vf->part(k).views);
				setFeaturesGradFromViews(vf->centres(), views.size() == 1 ? views[0] : torch::cat(views, 1));
				finishAdamGroup(1);
			}
			markWait(2);
			vf->reduction().waitAll();                    // ONE collective for the four small tensors (a SUM) ...
			markWait(3);
			finishGeomAdam(vf->reduction().gradScale());   // ... and one Adam launch that applies the 1/N
		} else {
			// each tensor is updated as soon as ITS reduction has landed (largest first)
			for (int i : red->order()) {
				red->wait(i);
				finishAdamGroup(i);
			}
		}
	} else {
		wait_all();
	}
	finishEnd();
	markWait(5);
	return loss;
}
