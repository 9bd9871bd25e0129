// Tensor<T>: shape + shared storage.  Same public surface as the reference's include/tensor.hpp
// (constructor, shape(), device(), data(), copy_from(), get_total_size(), DeviceType, TensorDataType,
// DEFAULT_ALLOC_METHOD, TensorFloat / TensorInt / TensorFloatPoint; copy construction aliases the
// storage, HOST tensors are pinned), re-implemented over a small type-erased HIP memory backend
// (memory.h) instead of one class hierarchy per allocation flavour.
#pragma once

#include <cstddef>
#include <memory>
#include <stdexcept>
#include <vector>

#include "memory.h"

enum class DeviceType { HOST, DEVICE };

enum class TensorDataType { SYNC_ALLOCATE = 0, ASYNC_ALLOCATE = 1 };

#ifdef DEFAULT_ALLOC_METHOD
#define MLI_ALLOC_METHOD_GIVEN 1
#else
#define DEFAULT_ALLOC_METHOD 0
#endif

constexpr TensorDataType DEFAULT_TENSOR_DATA_TYPE = static_cast<TensorDataType>(DEFAULT_ALLOC_METHOD);

#ifdef MLI_ALLOC_METHOD_GIVEN
// A host built with -DDEFAULT_ALLOC_METHOD=<n> (the reference's USE_ASYNC_ALLOC builds) makes <n> the flavour of every
// tensor in the process, the library's own included (memory.h: set_process_default_mode): registered before any tensor of
// this translation unit can be constructed.
namespace {
const bool mli_default_alloc_method_registered =
    (mli::mem::set_process_default_mode(DEFAULT_ALLOC_METHOD != 0 ? mli::mem::Mode::Async : mli::mem::Mode::Sync), true);
}
#endif

// the flavour of a tensor whose constructor is not told one
inline TensorDataType default_tensor_data_type() {
    return mli::mem::process_default_mode() == mli::mem::Mode::Async ? TensorDataType::ASYNC_ALLOCATE
                                                                     : TensorDataType::SYNC_ALLOCATE;
}

// Storage of one tensor.  The reference splits this into SyncTensorData / AsyncTensorData; here the
// flavour is a property of the backend block.
template <typename T>
class TensorData {
public:
    TensorData(std::size_t count, DeviceType device, TensorDataType flavour)
        : count_(count), device_(device),
          block_(mli::mem::acquire(count * sizeof(T),
                                   device == DeviceType::HOST ? mli::mem::Space::Host : mli::mem::Space::Device,
                                   flavour == TensorDataType::SYNC_ALLOCATE ? mli::mem::Mode::Sync
                                                                            : mli::mem::Mode::Async)) {}
    TensorData(const TensorData&) = delete;
    TensorData& operator=(const TensorData&) = delete;
    ~TensorData() { mli::mem::release(block_); }

    DeviceType device() const { return device_; }
    T* data() { return static_cast<T*>(mli::mem::pointer(block_)); }
    const T* data() const { return static_cast<const T*>(mli::mem::pointer(block_)); }

    void copy_from(const TensorData& other) {
        if (mli::mem::mode_of(other.block_) != mli::mem::mode_of(block_))
            throw std::runtime_error("Copy from: source uses a different allocation flavour");
        if (other.count_ != count_) throw std::runtime_error("Copy from: shape or device mismatch");
        mli::mem::copy(block_, other.block_, 0, count_ * sizeof(T));
    }

    // Extension (not in the reference): copy elements [first, first + count) only -- lets the scheduler
    // upload just the batch rows that changed instead of the whole inp[B, S] tensor.
    void copy_range_from(const TensorData& other, std::size_t first, std::size_t count) {
        if (mli::mem::mode_of(other.block_) != mli::mem::mode_of(block_))
            throw std::runtime_error("Copy from: source uses a different allocation flavour");
        if (other.count_ != count_ || first + count > count_)
            throw std::runtime_error("Copy from: shape or device mismatch");
        mli::mem::copy(block_, other.block_, first * sizeof(T), count * sizeof(T));
    }

    // Extension: element index[i] = value[i] for a handful of 8-byte elements (page-table entries) without
    // re-uploading the tensor.
    void scatter_from_host(const long long* index, const T* value, std::size_t n) {
        static_assert(sizeof(T) == 8 || sizeof(T) == 4, "scatter_from_host handles 4- and 8-byte elements");
        for (std::size_t i = 0; i < n; ++i)
            if (index[i] < 0 || static_cast<std::size_t>(index[i]) >= count_)
                throw std::runtime_error("scatter_from_host: index out of range");
        if constexpr (sizeof(T) == 8)
            mli::mem::scatter8(block_, index, reinterpret_cast<const unsigned long long*>(value), n);
        else
            mli::mem::scatter4(block_, index, reinterpret_cast<const unsigned int*>(value), n);
    }

    // Extension: stream-ordered copy of elements [first, first + count) that does not block the host (memory.h
    // copy_async: pinned host side, source unchanged until a later marker has been waited for).
    void copy_range_from_async(const TensorData& other, std::size_t first, std::size_t count) {
        if (other.count_ != count_ || first + count > count_)
            throw std::runtime_error("Copy from: shape or device mismatch");
        mli::mem::copy_async(block_, other.block_, first * sizeof(T), count * sizeof(T));
    }

private:
    std::size_t count_;
    DeviceType device_;
    mli::mem::Block* block_;
};

template <typename T>
class Tensor {
public:
    Tensor(const std::vector<std::size_t>& shape, DeviceType device = DeviceType::HOST,
           TensorDataType tensor_data_type = default_tensor_data_type())
        : shape_(shape), size_(1), device_(device) {
        for (std::size_t d : shape_) size_ *= d;
        data_ = std::make_shared<TensorData<T>>(size_, device, tensor_data_type);
    }
    Tensor(const Tensor&) = default;             // shallow: both tensors see the same memory
    Tensor& operator=(const Tensor&) = default;
    ~Tensor() = default;

    const std::vector<std::size_t>& shape() const { return shape_; }
    DeviceType device() const { return device_; }
    // Fetch the pointer once outside hot loops (the async flavour may wait on an event here).
    T* data() { return data_->data(); }
    const T* data() const { return data_->data(); }
    void copy_from(const Tensor& other) { data_->copy_from(*other.data_); }
    void copy_range_from(const Tensor& other, std::size_t first, std::size_t count) {
        data_->copy_range_from(*other.data_, first, count);
    }
    void scatter_from_host(const long long* index, const T* value, std::size_t n) {
        data_->scatter_from_host(index, value, n);
    }
    void copy_range_from_async(const Tensor& other, std::size_t first, std::size_t count) {
        data_->copy_range_from_async(*other.data_, first, count);
    }
    std::size_t get_total_size() const { return size_; }

private:
    std::vector<std::size_t> shape_;
    std::size_t size_;
    DeviceType device_;
    std::shared_ptr<TensorData<T>> data_;
};

typedef Tensor<float> TensorFloat;
typedef Tensor<int> TensorInt;
typedef Tensor<float*> TensorFloatPoint;
