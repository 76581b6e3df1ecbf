// Bookkeeping of the observed points and preference tuples
// (reference surface: include/sequential-line-search/preference-data-manager.hpp:11-43).
#ifndef SEQUENTIAL_LINE_SEARCH_PREFERENCE_DATA_MANAGER_HPP
#define SEQUENTIAL_LINE_SEARCH_PREFERENCE_DATA_MANAGER_HPP

#include <sequential-line-search/eigen-lite.hpp>
#include <sequential-line-search/preference.hpp>
#include <vector>

namespace sequential_line_search
{
    class PreferenceDataManager
    {
    public:
        /// Appends x_preferable and xs_other as new points plus the tuple "x_preferable beats xs_other"; optionally merges
        /// points closer than epsilon (the merged point is the midpoint and moves to the end of X).
        void AddNewPoints(const Eigen::VectorXd& x_preferable, const std::vector<Eigen::VectorXd>& xs_other,
                          const bool merge_close_points = true, const double epsilon = 1e-04);

        const Eigen::VectorXd GetLastSelectedDataPoint() const { return eig::Col(m_X, GetLastDataSample()[0]); }
        const Preference&     GetLastDataSample() const { return m_D.back(); }
        int                   GetNumDataPoints() const { return m_X.cols(); }
        const Eigen::MatrixXd&         GetX() const { return m_X; }
        const std::vector<Preference>& GetD() const { return m_D; }

    private:
        Eigen::MatrixXd         m_X;
        std::vector<Preference> m_D;
    };
} // namespace sequential_line_search

#endif
