// Device selection for the MI355X back end (extension: the reference is CPU-only and has no counterpart).
#ifndef SEQUENTIAL_LINE_SEARCH_DEVICE_HPP
#define SEQUENTIAL_LINE_SEARCH_DEVICE_HPP

#include <vector>

namespace sequential_line_search
{
    namespace device
    {
        /// GPUs the multi-start acquisition maximiser (acquisition_func::FindNextPoint*) shards its starts over.
        /// Default: $SLS_DEVICES ("0,1,2,3") or, if unset, the primary device $SLS_DEVICE (default 0) alone.
        /// Regressors are always fitted on the primary device; with more than one entry here the fitted state is replicated
        /// per device for the search and the per-device winners are merged with a single ncclAllGather.  A device may be
        /// listed more than once (logical shards on one GPU; the merge then happens on the host).
        void                    SetDevices(const std::vector<int>& devices);
        std::vector<int> Devices();   // by value: a copy taken under the lock (SetDevices may run on another thread)
    } // namespace device
} // namespace sequential_line_search

#endif
