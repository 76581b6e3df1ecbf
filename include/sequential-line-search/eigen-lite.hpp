// Minimal stand-in for the subset of Eigen the public surface of sequential-line-search uses
// (Eigen::MatrixXd / Eigen::VectorXd, column-major, value semantics).  Eigen itself is not installed in this image;
// when <Eigen/Core> IS available the real library is used instead and this file is inert.
#ifndef SEQUENTIAL_LINE_SEARCH_EIGEN_LITE_HPP
#define SEQUENTIAL_LINE_SEARCH_EIGEN_LITE_HPP

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(SLS_FORCE_EIGEN_LITE)
#include <Eigen/Cholesky>
#include <Eigen/Core>
#define SLS_HAVE_REAL_EIGEN 1
#endif
#endif

#ifndef SLS_HAVE_REAL_EIGEN

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <ostream>
#include <vector>

namespace Eigen
{
    using Index = long;

    namespace lite
    {
        /// Process-wide SplitMix64 stream behind Random().  Eigen proper draws from std::rand; that global stream is
        /// also consumed by other threads of the process (observed with the HIP runtime loaded: run-to-run different
        /// start sets), so the stand-in owns its generator.  Seed with sequential_line_search::utils::SetRandomSeed.
        inline unsigned long long& RandomState()
        {
            static unsigned long long state = 0x853c49e6748fea9bULL;
            return state;
        }
        inline double RandomUnit()
        {
            unsigned long long z = (RandomState() += 0x9E3779B97F4A7C15ULL);
            z                    = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z                    = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);
        }
    } // namespace lite

    class VectorXd
    {
    public:
        VectorXd() {}
        explicit VectorXd(Index n) : m_v(static_cast<size_t>(n), 0.0) {}
        VectorXd(std::initializer_list<double> l) : m_v(l) {}

        static VectorXd Zero(Index n) { return VectorXd(n); }
        static VectorXd Constant(Index n, double v)
        {
            VectorXd x(n);
            std::fill(x.m_v.begin(), x.m_v.end(), v);
            return x;
        }
        static VectorXd Ones(Index n) { return Constant(n, 1.0); }
        /// Uniform in [-1, 1] (Eigen's DenseBase::Random range) from the stand-in's own seeded stream.
        static VectorXd Random(Index n)
        {
            VectorXd x(n);
            for (auto& v : x.m_v) v = 2.0 * lite::RandomUnit() - 1.0;
            return x;
        }

        Index size() const { return static_cast<Index>(m_v.size()); }
        Index rows() const { return size(); }
        Index cols() const { return 1; }
        double*       data() { return m_v.data(); }
        const double* data() const { return m_v.data(); }
        double&       operator()(Index i) { return m_v[static_cast<size_t>(i)]; }
        double        operator()(Index i) const { return m_v[static_cast<size_t>(i)]; }
        double&       operator[](Index i) { return m_v[static_cast<size_t>(i)]; }
        double        operator[](Index i) const { return m_v[static_cast<size_t>(i)]; }

        VectorXd segment(Index start, Index n) const
        {
            VectorXd x(n);
            std::copy(m_v.begin() + start, m_v.begin() + start + n, x.m_v.begin());
            return x;
        }
        void setSegment(Index start, const VectorXd& s) { std::copy(s.m_v.begin(), s.m_v.end(), m_v.begin() + start); }

        double dot(const VectorXd& o) const
        {
            double s = 0.0;
            for (size_t i = 0; i < m_v.size(); ++i) s += m_v[i] * o.m_v[i];
            return s;
        }
        double squaredNorm() const { return dot(*this); }
        double norm() const { return std::sqrt(squaredNorm()); }
        double maxCoeff(int* index = nullptr) const
        {
            assert(!m_v.empty());
            size_t best = 0;
            for (size_t i = 1; i < m_v.size(); ++i)
                if (m_v[i] > m_v[best]) best = i;   // first maximum
            if (index) *index = static_cast<int>(best);
            return m_v[best];
        }
        VectorXd cwiseMax(const VectorXd& o) const
        {
            VectorXd x(size());
            for (size_t i = 0; i < m_v.size(); ++i) x.m_v[i] = std::max(m_v[i], o.m_v[i]);
            return x;
        }
        VectorXd cwiseMin(const VectorXd& o) const
        {
            VectorXd x(size());
            for (size_t i = 0; i < m_v.size(); ++i) x.m_v[i] = std::min(m_v[i], o.m_v[i]);
            return x;
        }
        const VectorXd& transpose() const { return *this; }

        VectorXd& operator+=(const VectorXd& o)
        {
            for (size_t i = 0; i < m_v.size(); ++i) m_v[i] += o.m_v[i];
            return *this;
        }
        VectorXd& operator-=(const VectorXd& o)
        {
            for (size_t i = 0; i < m_v.size(); ++i) m_v[i] -= o.m_v[i];
            return *this;
        }
        VectorXd& operator*=(double s)
        {
            for (auto& v : m_v) v *= s;
            return *this;
        }

    private:
        std::vector<double> m_v;
    };

    inline VectorXd operator+(VectorXd a, const VectorXd& b) { return a += b; }
    inline VectorXd operator-(VectorXd a, const VectorXd& b) { return a -= b; }
    inline VectorXd operator*(double s, VectorXd a) { return a *= s; }
    inline VectorXd operator*(VectorXd a, double s) { return a *= s; }
    inline VectorXd operator-(VectorXd a) { return a *= -1.0; }
    inline std::ostream& operator<<(std::ostream& os, const VectorXd& v)
    {
        for (Index i = 0; i < v.size(); ++i) os << (i ? " " : "") << v(i);
        return os;
    }

    class MatrixXd
    {
    public:
        MatrixXd() : m_r(0), m_c(0) {}
        MatrixXd(Index r, Index c) : m_r(r), m_c(c), m_v(static_cast<size_t>(r * c), 0.0) {}
        /// A vector is an n x 1 matrix (the 1-D demo assigns `X = x`).
        MatrixXd(const VectorXd& v) : m_r(v.size()), m_c(1), m_v(v.data(), v.data() + v.size()) {}

        static MatrixXd Zero(Index r, Index c) { return MatrixXd(r, c); }
        static MatrixXd Identity(Index r, Index c)
        {
            MatrixXd m(r, c);
            for (Index i = 0; i < std::min(r, c); ++i) m(i, i) = 1.0;
            return m;
        }

        Index rows() const { return m_r; }
        Index cols() const { return m_c; }
        Index size() const { return m_r * m_c; }
        double*       data() { return m_v.data(); }
        const double* data() const { return m_v.data(); }
        double&       operator()(Index i, Index j) { return m_v[static_cast<size_t>(i + j * m_r)]; }
        double        operator()(Index i, Index j) const { return m_v[static_cast<size_t>(i + j * m_r)]; }

        VectorXd col(Index j) const
        {
            VectorXd x(m_r);
            std::copy(m_v.begin() + j * m_r, m_v.begin() + (j + 1) * m_r, x.data());
            return x;
        }
        void setCol(Index j, const VectorXd& x) { std::copy(x.data(), x.data() + m_r, m_v.begin() + j * m_r); }
        /// Copy with one more column appended.
        MatrixXd withAppendedCol(const VectorXd& x) const
        {
            MatrixXd m(m_r == 0 ? x.size() : m_r, m_c + 1);
            std::copy(m_v.begin(), m_v.end(), m.m_v.begin());
            m.setCol(m_c, x);
            return m;
        }
        MatrixXd transpose() const
        {
            MatrixXd t(m_c, m_r);
            for (Index j = 0; j < m_c; ++j)
                for (Index i = 0; i < m_r; ++i) t(j, i) = (*this)(i, j);
            return t;
        }

    private:
        Index               m_r, m_c;
        std::vector<double> m_v;
    };
} // namespace Eigen

#endif // !SLS_HAVE_REAL_EIGEN

namespace sequential_line_search
{
    namespace eig
    {
        /// Column j of X as a vector (works for both the real Eigen and the stand-in).
        inline Eigen::VectorXd Col(const Eigen::MatrixXd& X, long j)
        {
            Eigen::VectorXd x(X.rows());
            for (long i = 0; i < X.rows(); ++i) x(i) = X(i, j);
            return x;
        }
        inline void SetCol(Eigen::MatrixXd& X, long j, const Eigen::VectorXd& x)
        {
            for (long i = 0; i < X.rows(); ++i) X(i, j) = x(i);
        }
        inline Eigen::MatrixXd AppendCol(const Eigen::MatrixXd& X, const Eigen::VectorXd& x)
        {
            Eigen::MatrixXd Y(x.size(), X.cols() + 1);
            for (long j = 0; j < X.cols(); ++j)
                for (long i = 0; i < X.rows(); ++i) Y(i, j) = X(i, j);
            for (long i = 0; i < x.size(); ++i) Y(i, X.cols()) = x(i);
            return Y;
        }
    } // namespace eig
} // namespace sequential_line_search

#endif // SEQUENTIAL_LINE_SEARCH_EIGEN_LITE_HPP
