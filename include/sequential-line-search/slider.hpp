// One-dimensional slider subspace (reference surface: include/sequential-line-search/slider.hpp:8-36).
#ifndef SEQUENTIAL_LINE_SEARCH_SLIDER_HPP
#define SEQUENTIAL_LINE_SEARCH_SLIDER_HPP

#include <sequential-line-search/eigen-lite.hpp>

namespace sequential_line_search
{
    class Slider
    {
    public:
        /// end_0 is expected to be x^+, end_1 the acquisition maximiser.  With `enlarge` the segment is stretched about
        /// its centre by `scale` as far as the [0,1]^D box allows, and to at least `minimum_length`.
        Slider(const Eigen::VectorXd& end_0, const Eigen::VectorXd& end_1, const bool enlarge, const double scale = 1.25,
               const double minimum_length = 0.25);

        Eigen::VectorXd GetValue(const double t) const { return (1.0 - t) * end_0 + t * end_1; }

        Eigen::VectorXd end_0;
        Eigen::VectorXd end_1;
        Eigen::VectorXd original_end_0;
        Eigen::VectorXd original_end_1;
    };
} // namespace sequential_line_search

#endif
