// PreferenceDataManager (reference: src/preference-data-manager.cpp).
#include <sequential-line-search/preference-data-manager.hpp>

using Eigen::MatrixXd;
using Eigen::VectorXd;

namespace sequential_line_search
{
    namespace
    {
        /// One pass: merge the first pair (i < j, row-major scan order) closer than epsilon; returns false if none.
        /// Semantics of src/preference-data-manager.cpp:14-86: the merged point is the midpoint, it becomes the LAST
        /// column, the other points keep their relative order, preference indices are remapped.
        bool MergeOnePair(const double eps_squared, MatrixXd& X, std::vector<Preference>& D)
        {
            const long M = X.cols();
            for (long i = 0; i < M; ++i)
                for (long j = i + 1; j < M; ++j)
                {
                    if ((eig::Col(X, i) - eig::Col(X, j)).squaredNorm() >= eps_squared) continue;
                    std::vector<unsigned> map(M);
                    unsigned              next = 0;
                    for (long k = 0; k < M; ++k)
                        if (k != i && k != j) map[k] = next++;
                    map[i] = map[j] = static_cast<unsigned>(M - 2);
                    MatrixXd Y(X.rows(), M - 1);
                    for (long k = 0; k < M; ++k)
                        if (k != i && k != j) eig::SetCol(Y, map[k], eig::Col(X, k));
                    eig::SetCol(Y, M - 2, 0.5 * (eig::Col(X, i) + eig::Col(X, j)));
                    X = Y;
                    for (Preference& p : D)
                        for (unsigned& idx : p) idx = map[idx];
                    return true;
                }
            return false;
        }
    } // namespace

    // reference: src/preference-data-manager.cpp:88-141
    void PreferenceDataManager::AddNewPoints(const VectorXd& x_preferable, const std::vector<VectorXd>& xs_other,
                                             const bool merge_close_points, const double epsilon)
    {
        const bool     first = (m_X.rows() == 0);
        const unsigned base  = first ? 0u : static_cast<unsigned>(m_X.cols());
        MatrixXd       X     = first ? MatrixXd(x_preferable.size(), 0) : m_X;
        X                    = eig::AppendCol(X, x_preferable);
        for (const VectorXd& x : xs_other) X = eig::AppendCol(X, x);
        m_X = X;

        std::vector<unsigned> indices(xs_other.size() + 1);
        for (unsigned i = 0; i < indices.size(); ++i) indices[i] = base + i;
        m_D.push_back(Preference(indices));

        if (!first && merge_close_points)   // the reference returns before merging on the very first call
            while (MergeOnePair(epsilon * epsilon, m_X, m_D)) {}
    }
} // namespace sequential_line_search
