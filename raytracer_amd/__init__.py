"""raytracer_amd — MI355X-native closest-hit / any-hit ray queries and radiosity (form factors + solve)
behind the C ABI of include/rtmi.h.  See DESIGN.md."""
from .api import (BuildScene, check_hierarchy, Camera, Radiosity, Scene, Shader, RtmiError, tone_map_dev, write_ppm, make_rays, triangle_intersection, primary_rays_dev, occlusion_fraction_dev, soft_shadow_factor_dev, jacobi_sweep_dev, lib,  # noqa: F401
                  RAY_DTYPE, HIT_DTYPE, RASTER_NORMAL, RASTER_ANTI_ALIASING, RASTER_DEPTH_OF_FIELD, MISS, INFINITY, FF_LIGHT_MEAN, SYMBOLS, LIB_PATH)
