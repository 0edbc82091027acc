// keyframe_batch_exchange.h -- the gradient exchange of a keyframe batch, one keyframe per rank (DESIGN.md
// section 6), on c10d::ProcessGroup: RCCL (ProcessGroupNCCL on ROCm) over xGMI on the GPU boxes, gloo in the host tests.
//
// The reference trains on one keyframe per step on one GPU (src/gaussian_mapper.cpp:620,677) and has no collective anywhere;
// this is the data-parallel extension of its train step, in the reference's host language.  The Python classes of the same
// names (photo-slam_amd/trainer.py: ViewFactoredExchange, GradientReduction) are mirrors for the tests.
//
//   ViewFactoredExchange   81 % of the gradient is the [P,16,3] SH tensor, and ONE view's SH gradient is rank one per
//                          Gaussian: basis(dir) x dL_dcolor with dir known to every rank.  So the ranks ALL-GATHER the 3-float
//                          colour gradients with the camera centre as one more row (ONE collective), each rebuilds the
//                          batch-mean SH gradient locally and applies it (TrainStep::stepFeaturesFromViews), and only the
//                          other four tensors (11 floats per Gaussian, ONE buffer, one more collective) are all-reduced.
//   GradientReduction      the plain mean of every leaf gradient, largest first, each tensor's Adam as soon as ITS reduction
//                          has landed.
//
// Work handles are waited on the compute stream (Work::wait() of ProcessGroupNCCL blocks the current stream, not the host):
// the host keeps queueing.  The process group comes from the caller: a C++ mapper builds c10d::TCPStore + ProcessGroupNCCL
// itself; under Python it is the default group, resolved by name (c10d::resolve_process_group).
#pragma once
#include <torch/torch.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <vector>

// a flat view over the storage the given contiguous fp32 tensors tile exactly and without gaps, or an undefined tensor
torch::Tensor oneBuffer(const std::vector<torch::Tensor>& tensors);

class GradientReduction {
public:
	// tensors: gradients, averaged in place over the ranks; members of one buffer (oneBuffer) share ONE collective.
	// sum_only: the collective SUMS and nothing scales -- the consumer multiplies by gradScale() = 1/N as it reads the
	// gradient (gsr_adam_multi_tensor.grad_scale): no averaging pass over the buffer, on RCCL (ncclAvg = pre-multiply +
	// sum; with ONE rank a whole extra kernel) as on gloo
	GradientReduction(c10::intrusive_ptr<c10d::ProcessGroup> pg, std::vector<torch::Tensor> tensors, bool sum_only = false);
	float gradScale() const { return sum_only_ ? 1.0f / static_cast<float>(pg_->getSize()) : 1.0f; }
	const std::vector<int>& order() const { return order_; }   // indices by decreasing size: the order the collectives were issued in
	void wait(int i);                                           // tensor i is reduced (stream-side on RCCL)
	void waitAll();
	int collectives() const { return static_cast<int>(groups_.size()); }

private:
	struct Group {
		torch::Tensor flat;
		c10::intrusive_ptr<c10d::Work> work;
		bool scaled = false;
	};
	c10::intrusive_ptr<c10d::ProcessGroup> pg_;
	std::vector<torch::Tensor> tensors_;
	std::vector<int> order_, group_of_;
	std::vector<Group> groups_;
	bool avg_ = false;   // the backend averages inside the collective (ncclAvg); gloo sums and wait() scales
	bool sum_only_ = false;
};

class ViewFactoredExchange {
public:
	// send: [P + 1, 3] (rows 0 .. P-1 = this view's colour gradients; row P receives the camera centre), camera_center [3],
	// others = the gradients of xyz / opacity / scaling / rotation.  After construction everything is in flight (two
	// collectives; round 2 sent the centres and two row ranges separately: four).
	// gather_stream (a hipStream_t; null = the current stream): the all-gather is issued with THAT stream current, so it waits
	// for whatever the stream waits for -- TrainStep hands over the stream gsr_backward made wait for "dL_dcolor_view is
	// complete", and the gather overlaps the last kernel of the backward pass.
	// gathered (optional): the [N, P + 1, 3] buffer the all-gather writes, allocated by the caller BEFORE the backward pass was
	// enqueued (TrainStep::sh_gathered_); without it one is allocated here, and a gather on a second stream is then made to wait
	// for the compute stream's tail first (a fresh block of the caching allocator may still be written by the pass).
	ViewFactoredExchange(c10::intrusive_ptr<c10d::ProcessGroup> pg, torch::Tensor send, torch::Tensor camera_center,
	                     std::vector<torch::Tensor> others, void* gather_stream = nullptr, torch::Tensor gathered = torch::Tensor());
	// The PACKED form (include/gsr.h: gsr_pack_color_view): every rank sends the rows its view SEES.  color_view: this view's
	// [P,3] colour gradient; capacity: rows a message holds -- max over the ranks of their views' visible counts, agreed
	// beforehand (TrainStep exchanges the counts right behind the forward pass); send / gathered: persistent int32 buffers of
	// packedViewWords(P, P rounded up) and N times that; the message is built and the gather issued with gather_stream current.
	struct Packed {
		torch::Tensor color_view, send, gathered, scratch;
		int64_t capacity = 0;
		bool prepacked = false;   // the backward pass has written the message into `send` (ShAdamStep::packed_view): nothing to pack
	};
	ViewFactoredExchange(c10::intrusive_ptr<c10d::ProcessGroup> pg, Packed packed, torch::Tensor camera_center,
	                     std::vector<torch::Tensor> others, void* gather_stream = nullptr);
	struct Part {
		int64_t row0 = 0;
		torch::Tensor views;   // [N, rows, 3]
		torch::Tensor messages;   // packed form: int32 [N * msg_stride], the N messages
		int64_t msg_stride = 0;   // words between two messages (0 = the dense form above)
		int64_t capacity = 0;     // rows every message has room for (packed form)
		c10::intrusive_ptr<c10d::Work> work;
	};
	int parts() const { return static_cast<int>(parts_.size()); }
	// part k once ITS all-gather has landed (stream-side wait): (first row, camera centres [N,3], colour gradients [N,rows,3])
	const Part& part(int k);
	torch::Tensor centres();
	GradientReduction& reduction() { return *reduction_; }
	void waitAll();

private:
	c10::intrusive_ptr<c10d::ProcessGroup> pg_;
	torch::Tensor gathered_, centres_;
	std::vector<Part> parts_;
	std::unique_ptr<GradientReduction> reduction_;
};
