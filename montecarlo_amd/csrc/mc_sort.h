// Internal: the key sort behind mc_rank_build (csrc/mc_sort.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

// Bytes of device workspace mc_sort_keys needs for n keys (two key buffers + control tables).
size_t mc_sort_workspace_bytes(uint64_t n);

// Sorts n u64 keys (fp64 bit patterns of non-negative doubles: IEEE bit order = numeric order) from `in`
// (left untouched) into `out`, ascending, asynchronously on `stream`.  Allocates nothing.  *err_word_out
// receives the device address of the sort's error word: non-zero after the sort means a table overflowed
// and `out` is not sorted (the caller poisons what it derives from it).  flags: MC_SORT_DEBUG_FAIL forces that
// failure path (tests); MC_SORT_FEW_REPEATS says no value is expected to repeat often (the search for heavy
// values and the kernel that writes their runs are not launched; any input is still sorted correctly).
// Inputs of 2^26 keys and more look for their heavy values in a larger sample (up to 3072 instead of 1024 values);
// MC_SORT_LARGE_SEARCH asks for that from 2^22 keys on, MC_SORT_SMALL_SEARCH never (tests and measurements).
// MC_SORT_RAW_F64: `in` holds fp64 bit patterns of ANY sign (NaN excluded from any promise); they are sorted in
// numeric order (-0.0 and 0.0 as one value) and `out` receives their ORDER WORDS, mc_sort_order_word(bits), ascending.
// MC_SORT_FRESH_SEARCH: the small search runs even when the workspace holds the heavy values of its last sort of n
// keys (by default those are taken again: keys from the same source repeat the same values; see k_sniff).
enum { MC_SORT_DEBUG_FAIL = 1, MC_SORT_FEW_REPEATS = 2, MC_SORT_LARGE_SEARCH = 4, MC_SORT_SMALL_SEARCH = 8,
       MC_SORT_RAW_F64 = 16, MC_SORT_FRESH_SEARCH = 32 };

// fp64 bits -> a word whose unsigned order is the numeric order (-0.0 and 0.0 compare equal in Python: one word)
__host__ __device__ inline unsigned long long mc_sort_order_word(unsigned long long b) {
    if ((b << 1) == 0) b = 0;
    return (b >> 63) ? ~b : (b | (1ull << 63));
}
// forgets the list of heavy values the workspace may hold from its last sort of n keys (the next sort searches)
int mc_sort_forget(void *workspace, uint64_t n, hipStream_t stream);
int mc_sort_keys(const uint64_t *in, uint64_t n, uint64_t *out, void *workspace, size_t workspace_bytes,
                 uint32_t **err_word_out, int flags, hipStream_t stream);

// Host copy of the sort's counters after `stream` has drained: {error word, heavy values, tail keys,
// levels used, finish tiles, segments created, 0, 0}.
int mc_sort_info(const void *workspace, uint64_t n, uint64_t info[8], hipStream_t stream);
