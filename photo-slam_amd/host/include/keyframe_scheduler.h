// keyframe_scheduler.h -- which keyframe(s) a train step uses: the bookkeeping of
// GaussianMapper::useOneRandomSlidingWindowKeyframe (src/gaussian_mapper.cpp:1126-1173; generateKfidRandomShuffle :1103-1124,
// increaseKeyframeTimesOfUse :1199-1204) on keyframe INDICES (the position of a keyframe in scene_->keyframes(), a std::map
// ordered by frame id), so that it can serve a BATCH: a data-parallel step trains B = #ranks keyframes at once (SURVEY.md 8(e):
// "the reference scheduler called B times on rank 0 and broadcast as indices").
//
// The walk is the reference's: a random permutation of the indices, made again whenever a keyframe has been added (addKeyframe
// clears kfid_shuffled_, :438 / :1065); the cursor advances cyclically and skips keyframes whose remaining times of use are
// spent; a full cycle that finds none grants every keyframe one more use (:1148-1150); the chosen keyframe's use count goes up
// (kfs_used_times_, the step of its position learning rate: :663-671) and its remaining uses go down.  The permutation comes
// from std::mt19937 + std::shuffle like the reference's, seeded by the caller instead of std::random_device: every rank that
// constructs the scheduler with the same seed and feeds it the same calls holds the same state -- useBatchOnRanks() lets rank 0
// draw and the others replay its choices, so a diverged replica shows up as an exception, not as two ranks training one keyframe.
#pragma once
#include <cstdint>
#include <random>
#include <vector>

#include <torch/torch.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

class KeyframeScheduler {
public:
	explicit KeyframeScheduler(uint64_t seed) : rng_(static_cast<std::mt19937::result_type>(seed)) {}

	// a keyframe joins the session (GaussianScene::addKeyframe + increaseKeyframeTimesOfUse(pkf, newKeyframeTimesOfUse())): returns its index
	int addKeyframe(int times_of_use);
	void increaseTimesOfUse(int keyframe, int times);   // local BA / loop closure grant more (:841, :930)
	int size() const { return static_cast<int>(remaining_.size()); }
	int usedTimes(int keyframe) const { return used_.at(static_cast<size_t>(keyframe)); }
	int remainingTimesOfUse(int keyframe) const { return remaining_.at(static_cast<size_t>(keyframe)); }

	// useOneRandomSlidingWindowKeyframe: the next keyframe's index, -1 when the session has none
	int useOne();
	// B consecutive draws (one step's batch); an index may repeat when fewer than B keyframes have uses left
	std::vector<int> useBatch(int B);
	// the data-parallel form: rank 0 draws the batch of B = group size, broadcasts the indices, every other rank draws its own and
	// checks that it drew the same (throws otherwise: the replicas' sessions have diverged).  Rank r trains on result[r].
	std::vector<int> useBatchOnRanks(const c10::intrusive_ptr<c10d::ProcessGroup>& group);

private:
	void shuffle();
	std::mt19937 rng_;
	std::vector<int> remaining_, used_, order_;
	bool shuffled_ = false;
	size_t cursor_ = 0;
};
