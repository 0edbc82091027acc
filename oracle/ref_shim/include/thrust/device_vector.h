/* shim of the sliver of thrust::device_vector simple-knn uses (host memory is "device" memory here); test infrastructure */
#pragma once
#include <vector>

namespace thrust {
template <typename T>
struct device_ptr {
	T* p;
	T* get() const { return p; }
};
template <typename T>
class device_vector {
public:
	device_vector() {}
	explicit device_vector(size_t n) : v_(n) {}
	device_ptr<T> data() { return device_ptr<T>{v_.data()}; }
	typename std::vector<T>::iterator begin() { return v_.begin(); }
	typename std::vector<T>::iterator end() { return v_.end(); }
	size_t size() const { return v_.size(); }
	void resize(size_t n) { v_.resize(n); }
private:
	std::vector<T> v_;
};
}  // namespace thrust
