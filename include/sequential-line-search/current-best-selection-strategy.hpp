#ifndef SEQUENTIAL_LINE_SEARCH_CURRENT_BEST_SELECTION_STRATEGY_HPP
#define SEQUENTIAL_LINE_SEARCH_CURRENT_BEST_SELECTION_STRATEGY_HPP

namespace sequential_line_search
{
    /// How x^+ (the "current best" end of the next slider) is chosen.
    enum class CurrentBestSelectionStrategy
    {
        LargestExpectValue,
        LastSelection,
    };
} // namespace sequential_line_search

#endif
