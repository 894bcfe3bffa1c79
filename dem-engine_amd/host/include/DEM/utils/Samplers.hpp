// DEM/utils/Samplers.hpp -- point samplers for filling a volume with clump centres, with the class names and calling
// conventions of the reference (src/DEM/utils/Samplers.hpp: Sampler base with SampleBox / SampleSphere / SampleCylinderX|Y|Z;
// GridSampler, HCPSampler, PDSampler; DEMBoxGridSampler, DEMBoxHCPSampler, DEMCylSurfSampler).  Own implementations:
//  * GridSampler / HCPSampler generate the same lattices (origin at the lower corner of the bounding box; HCP rows dx/2 and
//    layers dy/3 apart as the reference lays them), so deterministic scripts place the same particles;
//  * PDSampler is a seeded dart-throwing Poisson-disk sampler on a background grid (Bridson's algorithm): the same guarantee
//    (no two points closer than the separation), not the same point set -- the reference seeds from std::random_device, so
//    its sets differ from run to run anyway.
#pragma once
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "../HostSideHelpers.hpp"

namespace deme {

enum class SamplingType { REGULAR_GRID, POISSON_DISK, HCP_PACK };

class Sampler {
  public:
    explicit Sampler(float separation) : m_separation(separation) {}
    virtual ~Sampler() {}

    std::vector<float3> SampleBox(const float3& center, const float3& halfDim) {
        m_center = center, m_size = halfDim;
        return Sample(BOX);
    }
    std::vector<float3> SampleSphere(const float3& center, float radius) {
        m_center = center, m_size = make_float3(radius, radius, radius);
        return Sample(SPHERE);
    }
    std::vector<float3> SampleCylinderX(const float3& center, float radius, float halfHeight) {
        m_center = center, m_size = make_float3(halfHeight, radius, radius);
        return Sample(CYLINDER_X);
    }
    std::vector<float3> SampleCylinderY(const float3& center, float radius, float halfHeight) {
        m_center = center, m_size = make_float3(radius, halfHeight, radius);
        return Sample(CYLINDER_Y);
    }
    std::vector<float3> SampleCylinderZ(const float3& center, float radius, float halfHeight) {
        m_center = center, m_size = make_float3(radius, radius, halfHeight);
        return Sample(CYLINDER_Z);
    }
    // std::vector<float> twins (the reference's Python-friendly forms)
    std::vector<std::vector<float>> SampleBox(const std::vector<float>& center, const std::vector<float>& halfDim) {
        return unpack(SampleBox(v3(center, "SampleBox"), v3(halfDim, "SampleBox")));
    }
    std::vector<std::vector<float>> SampleSphere(const std::vector<float>& center, float radius) {
        return unpack(SampleSphere(v3(center, "SampleSphere"), radius));
    }
    std::vector<std::vector<float>> SampleCylinderX(const std::vector<float>& center, float radius, float halfHeight) {
        return unpack(SampleCylinderX(v3(center, "SampleCylinderX"), radius, halfHeight));
    }
    std::vector<std::vector<float>> SampleCylinderY(const std::vector<float>& center, float radius, float halfHeight) {
        return unpack(SampleCylinderY(v3(center, "SampleCylinderY"), radius, halfHeight));
    }
    std::vector<std::vector<float>> SampleCylinderZ(const std::vector<float>& center, float radius, float halfHeight) {
        return unpack(SampleCylinderZ(v3(center, "SampleCylinderZ"), radius, halfHeight));
    }
    virtual float GetSeparation() const { return m_separation; }
    virtual void SetSeparation(float separation) { m_separation = separation; }

  protected:
    enum VolumeType { BOX, SPHERE, CYLINDER_X, CYLINDER_Y, CYLINDER_Z };
    virtual std::vector<float3> Sample(VolumeType t) = 0;

    /// is p inside the sampling volume (box faces and cylinder caps with a relative fuzz, like the reference)
    bool accept(VolumeType t, const float3& p) const {
        const float3 v = p - m_center;
        const float fuzz = (m_size.x < 1) ? 1e-6f * m_size.x : 1e-6f;
        switch (t) {
            case BOX:
                return std::abs(v.x) <= m_size.x + fuzz && std::abs(v.y) <= m_size.y + fuzz && std::abs(v.z) <= m_size.z + fuzz;
            case SPHERE:
                return dot(v, v) <= m_size.x * m_size.x;
            case CYLINDER_X:
                return v.y * v.y + v.z * v.z <= m_size.y * m_size.y && std::abs(v.x) <= m_size.x + fuzz;
            case CYLINDER_Y:
                return v.z * v.z + v.x * v.x <= m_size.z * m_size.z && std::abs(v.y) <= m_size.y + fuzz;
            case CYLINDER_Z:
                return v.x * v.x + v.y * v.y <= m_size.x * m_size.x && std::abs(v.z) <= m_size.z + fuzz;
        }
        return false;
    }
    static float3 v3(const std::vector<float>& v, const char* who) {
        if (v.size() != 3)
            throw std::runtime_error(std::string(who) + ": a 3-element vector is expected");
        return make_float3(v[0], v[1], v[2]);
    }
    static std::vector<std::vector<float>> unpack(const std::vector<float3>& pts) {
        std::vector<std::vector<float>> out(pts.size());
        for (size_t i = 0; i < pts.size(); i++)
            out[i] = {pts[i].x, pts[i].y, pts[i].z};
        return out;
    }

    float m_separation;
    float3 m_center{0, 0, 0};
    float3 m_size{0, 0, 0};  // half dimensions of the bounding box of the volume
};

/// regular grid, x-major outer loop as in the reference (Samplers.hpp:550-570)
class GridSampler : public Sampler {
  public:
    explicit GridSampler(float separation) : Sampler(separation), m_sep3D(make_float3(separation, separation, separation)) {}
    explicit GridSampler(const float3& separation) : Sampler(separation.x), m_sep3D(separation) {}
    void SetSeparation(float separation) override { m_sep3D = make_float3(separation, separation, separation); }

  private:
    std::vector<float3> Sample(VolumeType t) override {
        std::vector<float3> out;
        const float3 bl = m_center - m_size;
        const int nx = (int)(2 * m_size.x / m_sep3D.x) + 1, ny = (int)(2 * m_size.y / m_sep3D.y) + 1, nz = (int)(2 * m_size.z / m_sep3D.z) + 1;
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++)
                for (int k = 0; k < nz; k++) {
                    const float3 p = bl + make_float3(i * m_sep3D.x, j * m_sep3D.y, k * m_sep3D.z);
                    if (accept(t, p))
                        out.push_back(p);
                }
        return out;
    }
    float3 m_sep3D;
};

/// hexagonal close packing: rows sqrt(3)/2 apart, layers sqrt(2/3) apart, alternate rows shifted dx/2, alternate layers dy/3
class HCPSampler : public Sampler {
  public:
    explicit HCPSampler(float separation) : Sampler(separation) {}

  private:
    std::vector<float3> Sample(VolumeType t) override {
        std::vector<float3> out;
        const float3 bl = m_center - m_size;
        const float dx = m_separation, dy = m_separation * (float)(std::sqrt(3.0) / 2), dz = m_separation * (float)std::sqrt(2.0 / 3.0);
        const int nx = (int)(2 * m_size.x / dx) + 1, ny = (int)(2 * m_size.y / dy) + 1, nz = (int)(2 * m_size.z / dz) + 1;
        for (int k = 0; k < nz; k++) {
            const float oy = (k % 2 == 0) ? 0.f : dy / 3;
            for (int j = 0; j < ny; j++) {
                const float ox = ((j + k) % 2 == 0) ? 0.f : dx / 2;
                for (int i = 0; i < nx; i++) {
                    const float3 p = bl + make_float3(ox + i * dx, oy + j * dy, k * dz);
                    if (accept(t, p))
                        out.push_back(p);
                }
            }
        }
        return out;
    }
};

/// Poisson-disk sampling (no two points closer than the separation): Bridson's algorithm on a background grid of cell size
/// separation / sqrt(3), seeded (default seed fixed: reproducible scenes; pass another seed for another set).  A volume
/// thinner than the separation along z is sampled as a 2-D layer, like the reference does for its flat boxes.
class PDSampler : public Sampler {
  public:
    explicit PDSampler(float separation, int pointsPerIteration = 30, uint64_t seed = 20240928ull)
        : Sampler(separation), m_ppi(pointsPerIteration), m_rng(seed) {}
    void SetRandomEngineSeed(uint64_t seed) { m_rng.seed(seed); }

  private:
    std::vector<float3> Sample(VolumeType t) override {
        const float r = m_separation;
        const bool flat = 2 * m_size.z < r;  // one layer
        const float cell = r / std::sqrt(flat ? 2.f : 3.f);
        const float3 lo = m_center - m_size;
        const int gx = std::max(1, (int)std::ceil(2 * m_size.x / cell)), gy = std::max(1, (int)std::ceil(2 * m_size.y / cell)),
                  gz = flat ? 1 : std::max(1, (int)std::ceil(2 * m_size.z / cell));
        std::vector<int> grid((size_t)gx * gy * gz, -1);
        std::vector<float3> pts;
        std::vector<int> active;
        std::uniform_real_distribution<float> U(0.f, 1.f);
        auto cell_of = [&](const float3& p, int& i, int& j, int& k) {
            i = std::min(gx - 1, std::max(0, (int)((p.x - lo.x) / cell)));
            j = std::min(gy - 1, std::max(0, (int)((p.y - lo.y) / cell)));
            k = flat ? 0 : std::min(gz - 1, std::max(0, (int)((p.z - lo.z) / cell)));
        };
        auto far_enough = [&](const float3& p) {
            int i, j, k;
            cell_of(p, i, j, k);
            for (int c = std::max(0, k - 2); c <= std::min(gz - 1, k + 2); c++)
                for (int b = std::max(0, j - 2); b <= std::min(gy - 1, j + 2); b++)
                    for (int a = std::max(0, i - 2); a <= std::min(gx - 1, i + 2); a++) {
                        const int q = grid[((size_t)c * gy + b) * gx + a];
                        if (q >= 0 && dot(pts[q] - p, pts[q] - p) < r * r)
                            return false;
                    }
            return true;
        };
        auto add = [&](const float3& p) {
            int i, j, k;
            cell_of(p, i, j, k);
            grid[((size_t)k * gy + j) * gx + i] = (int)pts.size();
            active.push_back((int)pts.size());
            pts.push_back(p);
        };
        // first point: the centre (always inside); then grow from the active list
        add(flat ? make_float3(m_center.x, m_center.y, m_center.z) : m_center);
        while (!active.empty()) {
            const size_t pick = (size_t)(U(m_rng) * active.size()) % active.size();
            const float3 base = pts[active[pick]];
            bool found = false;
            for (int trial = 0; trial < m_ppi; trial++) {
                // uniform in the shell [r, 2r)
                const float rad = r * (1.f + U(m_rng));
                float3 d;
                if (flat) {
                    const float a = 6.2831853f * U(m_rng);
                    d = make_float3(std::cos(a), std::sin(a), 0.f);
                } else {
                    const float z = 2.f * U(m_rng) - 1.f, a = 6.2831853f * U(m_rng), s = std::sqrt(std::max(0.f, 1.f - z * z));
                    d = make_float3(s * std::cos(a), s * std::sin(a), z);
                }
                const float3 p = base + d * rad;
                if (accept(t, p) && far_enough(p)) {
                    add(p);
                    found = true;
                    break;
                }
            }
            if (!found) {
                active[pick] = active.back();
                active.pop_back();
            }
        }
        return pts;
    }
    int m_ppi;
    std::mt19937_64 m_rng;
};

inline std::vector<float3> DEMBoxGridSampler(float3 BoxCenter, float3 HalfDims, float GridSizeX, float GridSizeY = -1.0,
                                             float GridSizeZ = -1.0) {
    if (GridSizeY < 0)
        GridSizeY = GridSizeX;
    if (GridSizeZ < 0)
        GridSizeZ = GridSizeX;
    GridSampler sampler(make_float3(GridSizeX, GridSizeY, GridSizeZ));
    return sampler.SampleBox(BoxCenter, HalfDims);
}
inline std::vector<float3> DEMBoxHCPSampler(float3 BoxCenter, float3 HalfDims, float GridSize) {
    HCPSampler sampler(GridSize);
    return sampler.SampleBox(BoxCenter, HalfDims);
}
/// a shell of particles on a cylindrical surface: rows along the axis, spacing * ParticleRad apart
inline std::vector<float3> DEMCylSurfSampler(float3 CylCenter, float3 CylAxis, float CylRad, float CylHeight, float ParticleRad,
                                             float spacing = 1.2f) {
    std::vector<float3> points;
    const float perimeter = (float)(2.0 * PI * CylRad);
    const unsigned int rows = (unsigned int)(perimeter / (spacing * ParticleRad));
    const float dAngle = (float)(2.0 * PI / (double)rows), dSide = spacing * ParticleRad;
    const float3 axis = normalize(CylAxis);
    float3 radial = findPerpendicular<float3>(axis);
    for (unsigned int i = 0; i < rows; i++) {
        const float3 start = CylCenter + axis * (CylHeight / 2.f) + radial * CylRad;
        for (float d = 0.f; d <= CylHeight; d += dSide)
            points.push_back(start + axis * (-d));
        radial = Rodrigues(radial, axis, dAngle);
    }
    return points;
}

}  // namespace deme
