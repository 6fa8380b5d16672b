"""raytracing_amd -- MI355X (gfx950) render backend for the per-pixel trace loop of avneic/RayTracing.

The product is the C-ABI library ``librtx.so`` (``include/rtx.h``; sources in ``raytracing_amd/csrc``).  This package
is its Python host side: a ctypes binding plus a mirror of the reference's operator interface for the path
(``renderScene`` family, raytracer.h:17-19; ``Scene``/``Sphere``/``material_t``/``Camera``, scene.h, sphere.h,
material.h, camera.h).  There is no CPU fallback: without the built library, or without a gfx950 device, rendering
raises.
"""
from .api import (Camera, Context, Group, Material, RtxError, Scene, Sphere, Stats, book_camera, lib, random_scene,
                  render_scene_hip, write_ppm, plan_chunks, build_grid, spp_range, MATERIAL_DIFFUSE, MATERIAL_GLASS, MATERIAL_METAL, FLAG_ACCUMULATE, FLAG_RECURSIVE, FLAG_VALIDATE_FILTER,
                  FILTER_MFMA, FILTER_VALU, FILTER_GRID, FILTER_AUTO)

__all__ = ["Camera", "Context", "Group", "Material", "RtxError", "Scene", "Sphere", "Stats", "book_camera", "lib", "random_scene",
           "render_scene_hip", "write_ppm", "plan_chunks", "build_grid", "spp_range", "MATERIAL_DIFFUSE", "MATERIAL_GLASS", "MATERIAL_METAL", "FLAG_ACCUMULATE", "FLAG_RECURSIVE", "FLAG_VALIDATE_FILTER", "FILTER_MFMA", "FILTER_VALU", "FILTER_GRID", "FILTER_AUTO"]
