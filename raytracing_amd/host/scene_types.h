// scene_types.h -- the host-side scene-construction types of the standalone CLI (own code, same names and fields as
// the reference's scene.h:13-47, sphere.h:10-21, material.h:9-38 and camera.h:15-102 so that render_scene_hip.h
// compiles against either).  A reference build includes its own headers instead of this file.
#pragma once
#include "../../include/rtx.h"

#include <cfloat>
#include <vector>

namespace pk
{

struct vector3 {
    float x, y, z;
    vector3() : x(0), y(0), z(0) {}
    vector3(float a, float b, float c) : x(a), y(b), z(c) {}
};

typedef enum { MATERIAL_NONE = 0, MATERIAL_DIFFUSE = 1, MATERIAL_METAL = 2, MATERIAL_GLASS = 3 } material_type_t;

struct material_t {
    material_type_t type;
    vector3         albedo;
    float           blur;
    float           refractionIndex;
    material_t() : type(MATERIAL_NONE), albedo(1, 0, 1), blur(1.0f), refractionIndex(1.0f) {}
    material_t(material_type_t t, vector3 a = vector3(1, 0, 1), float b = 1.0f, float r = 1.0f) : type(t), albedo(a), blur(b), refractionIndex(r) {}
};

class IVisible {
public:
    virtual ~IVisible() {}
};

class Sphere : public IVisible {
public:
    Sphere(const vector3& pos, float r, material_t* m) : center(pos), radius(r), material(m) {}
    ~Sphere() override { delete material; }
    vector3     center;
    float       radius;
    material_t* material;
};

class Scene {
public:
    ~Scene() { for (IVisible* o : objects) delete o; }
    std::vector<IVisible*> objects;
};

// The derived members are filled by librtx's rtx_camera_init, which follows camera.h:20-41 operation by operation.
class Camera {
public:
    Camera(float vfov, float aspect, float aperture = 1.0f, float focusDistance = FLT_MAX, const vector3& pos = vector3(0, 0, 0),
           const vector3& up = vector3(0, 1, 0), const vector3& lookat = vector3(0, 0, -1))
    {
        const float p[3] = { pos.x, pos.y, pos.z }, u3[3] = { up.x, up.y, up.z }, l[3] = { lookat.x, lookat.y, lookat.z };
        rtx_camera c;
        rtx_camera_init(&c, vfov, aspect, aperture, focusDistance, p, u3, l);
        origin = vector3(c.origin[0], c.origin[1], c.origin[2]);
        leftCorner = vector3(c.left_corner[0], c.left_corner[1], c.left_corner[2]);
        horizontal = vector3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
        vertical = vector3(c.vertical[0], c.vertical[1], c.vertical[2]);
        u = vector3(c.u[0], c.u[1], c.u[2]);
        v = vector3(c.v[0], c.v[1], c.v[2]);
        w = vector3(c.w[0], c.w[1], c.w[2]);
        lensRadius = c.lens_radius;
    }
    vector3 origin, leftCorner, horizontal, vertical, u, v, w;
    float   lensRadius;
};

} // namespace pk
