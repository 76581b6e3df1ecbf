// Slider (reference: src/slider.cpp).  The reference finds the enlargement factors with two 1-D COBYLA solves
// (src/slider.cpp:72-97); the constraint set is an interval, so the solution is available in closed form:
// the largest t <= scale with crop(c +/- t r) == c +/- t r.
#include <cmath>
#include <sequential-line-search/slider.hpp>
#include <utility>

using Eigen::VectorXd;

namespace sequential_line_search
{
    namespace
    {
        const double kEps = 1e-16;   // src/slider.cpp:19-23

        double Crop(double x) { return x > kEps ? (x < 1.0 - kEps ? x : 1.0 - kEps) : kEps; }
        VectorXd Crop(const VectorXd& x)
        {
            VectorXd y(x.size());
            for (long i = 0; i < x.size(); ++i) y(i) = Crop(x(i));
            return y;
        }

        /// Largest t in [0, scale] such that c + t * dir stays inside [eps, 1 - eps]^D.
        double MaxFeasibleStep(const VectorXd& c, const VectorXd& dir, double scale)
        {
            double t = scale;
            for (long i = 0; i < c.size(); ++i)
            {
                if (dir(i) > 0.0) t = std::min(t, (1.0 - kEps - c(i)) / dir(i));
                else if (dir(i) < 0.0) t = std::min(t, (kEps - c(i)) / dir(i));
            }
            return std::max(t, 0.0);
        }

        std::pair<VectorXd, VectorXd> Enlarge(const VectorXd& x_1, const VectorXd& x_2, double scale, double minimum_length)
        {
            const VectorXd c = 0.5 * (Crop(x_1) + Crop(x_2));
            const VectorXd r = Crop(x_1) - c;
            const double t_1 = MaxFeasibleStep(c, r, scale);
            const double t_2 = MaxFeasibleStep(c, -r, scale);

            const VectorXd e_1 = Crop(c + t_1 * r);
            const VectorXd e_2 = Crop(c - t_2 * r);
            const double   len = (e_1 - e_2).norm();
            if (len < minimum_length)   // src/slider.cpp:104-118
            {
                const double k = minimum_length / len;
                if (std::abs(t_1 - t_2) < 1e-10) return {c + k * t_1 * r, c - k * t_2 * r};
                if (t_1 > t_2) return {c + 2.0 * k * t_1 * r, c - t_2 * r};
                return {c + t_1 * r, c - 2.0 * k * t_2 * r};
            }
            return {e_1, e_2};
        }
    } // namespace

    Slider::Slider(const VectorXd& end_0_, const VectorXd& end_1_, const bool enlarge, const double scale, const double minimum_length)
        : original_end_0(end_0_), original_end_1(end_1_)
    {
        if (enlarge)
        {
            const auto ends = Enlarge(original_end_0, original_end_1, scale, minimum_length);
            end_0           = ends.first;
            end_1           = ends.second;
        }
        else
        {
            end_0 = original_end_0;
            end_1 = original_end_1;
        }
    }
} // namespace sequential_line_search
