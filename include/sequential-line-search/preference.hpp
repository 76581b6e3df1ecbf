// One preferential observation among 2+ data points (reference: include/sequential-line-search/preference.hpp:8-21).
#ifndef SEQUENTIAL_LINE_SEARCH_PREFERENCE_HPP
#define SEQUENTIAL_LINE_SEARCH_PREFERENCE_HPP

#include <vector>

namespace sequential_line_search
{
    /// Index list whose FIRST element is the preferred data point.
    struct Preference : public std::vector<unsigned>
    {
        Preference(unsigned i, unsigned j) : std::vector<unsigned>{i, j} {}
        Preference(unsigned i, unsigned j, unsigned k) : std::vector<unsigned>{i, j, k} {}
        Preference(const std::vector<unsigned>& indices) : std::vector<unsigned>{indices} {}
    };
} // namespace sequential_line_search

#endif
