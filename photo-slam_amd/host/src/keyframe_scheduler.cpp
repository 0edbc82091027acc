// keyframe_scheduler.cpp -- see host/include/keyframe_scheduler.h
#include "keyframe_scheduler.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>

int KeyframeScheduler::addKeyframe(int times_of_use)
{
	remaining_.push_back(times_of_use);
	used_.push_back(0);
	shuffled_ = false;   // (GaussianScene::addKeyframe(pkf, &kfid_shuffled_): the permutation is made again before the next draw)
	return size() - 1;
}

void KeyframeScheduler::increaseTimesOfUse(int keyframe, int times) { remaining_.at(static_cast<size_t>(keyframe)) += times; }

void KeyframeScheduler::shuffle()
{
	order_.resize(remaining_.size());
	std::iota(order_.begin(), order_.end(), 0);
	std::shuffle(order_.begin(), order_.end(), rng_);
	shuffled_ = true;
}

int KeyframeScheduler::useOne()
{
	if (remaining_.empty()) return -1;
	if (!shuffled_) shuffle();
	// (the cursor survives a reshuffle, as kfid_shuffle_idx_ does; it may then point beyond a permutation of another length only if
	// keyframes were removed, which this class does not do)
	const size_t start = cursor_;
	int chosen;
	do {
		if (++cursor_ >= order_.size()) cursor_ = 0;
		if (cursor_ == start)   // a whole cycle without a usable keyframe: every keyframe gets one more use
			for (int& r : remaining_) r += 1;
		chosen = order_[cursor_];
	} while (remaining_[static_cast<size_t>(chosen)] <= 0);
	used_[static_cast<size_t>(chosen)] += 1;
	remaining_[static_cast<size_t>(chosen)] -= 1;
	return chosen;
}

std::vector<int> KeyframeScheduler::useBatch(int B)
{
	std::vector<int> batch;
	for (int i = 0; i < B; i++) batch.push_back(useOne());
	return batch;
}

std::vector<int> KeyframeScheduler::useBatchOnRanks(const c10::intrusive_ptr<c10d::ProcessGroup>& group)
{
	const int B = group->getSize(), rank = group->getRank();
	const std::vector<int> mine = useBatch(B);   // every rank books the same draws: the state stays replicated without being exchanged
	auto host = torch::empty({B}, torch::kInt64);
	for (int i = 0; i < B; i++) host[i] = static_cast<int64_t>(mine[static_cast<size_t>(i)]);
	// (RCCL moves device tensors, gloo host tensors)
	auto t = group->getBackendName() == "nccl" ? host.to(torch::kCUDA) : host;
	std::vector<at::Tensor> tensors{t};
	c10d::BroadcastOptions opts;
	opts.rootRank = 0;
	group->broadcast(tensors, opts)->wait();
	host = t.to(torch::kCPU);
	std::vector<int> batch(static_cast<size_t>(B));
	for (int i = 0; i < B; i++) batch[static_cast<size_t>(i)] = static_cast<int>(host[i].item<int64_t>());
	if (rank != 0 && batch != mine)
		throw std::runtime_error("KeyframeScheduler::useBatchOnRanks: this rank drew another batch than rank 0 -- the replicas' keyframe "
		                         "sessions have diverged (different seed, or calls that did not reach every rank)");
	return batch;
}
