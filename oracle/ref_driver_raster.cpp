// TEST INFRASTRUCTURE — not part of the product (see ref_driver.cpp).
//
// RasterDetails, the reference's camera-basis struct, is defined inside
// rasterizer.cpp (Render2/rasterizer.cpp:11-69), not in a header.  To run the
// reference's own constructor we compile rasterizer.cpp *as part of this
// translation unit*, from where it lies; oracle/Makefile therefore does not
// compile rasterizer.cpp a second time.
#include "rasterizer.cpp" // found through -I/root/reference/Render2

// G,U,R,O,dR,dU (6 × xyz) exactly as RasterDetails' constructor computes them (rasterizer.cpp:49-56)
extern "C" void ref_raster_details_impl(Scene *scene, Camera *cam, double *out18) {
	string name = "unused";
	PPM_Image img(1, 1);
	RasterDetails rd(*scene, *cam, name, img);
	const Vec3 *v[6] = {&rd.G, &rd.U, &rd.R, &rd.O, &rd.dR, &rd.dU};
	for (int k = 0; k < 6; ++k) { out18[3 * k] = v[k]->x; out18[3 * k + 1] = v[k]->y; out18[3 * k + 2] = v[k]->z; }
}

// ToneMap (rasterizer.cpp:79-88, file-static) on n colours -> n × 3 bytes
extern "C" void ref_tone_map(long n, const double *rgb, unsigned char *out) {
	for (long k = 0; k < n; ++k) {
		Pixel p = ToneMap(Color(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]));
		out[3 * k] = p.r; out[3 * k + 1] = p.g; out[3 * k + 2] = p.b;
	}
}

// The reference's own Rasterizer::Normal_Raster / Anti_Aliasing / Depth_Of_Field_Effect (rasterizer.cpp:243-406) over
// the 8 × 6 thread rectangles they are written for (x_res must divide by 8, y_res by 6), into a PPM_Image:
// pixels = x_res·y_res·3 ToneMap bytes.  They also write <workdir>/<name>.ppm after every rectangle.  mode 0, 1, 2.
extern "C" int ref_raster_frame(Scene *scene, Camera *cam, int mode, const char *workdir, unsigned char *pixels) {
	if (cam->x_res % threadDivisionsInX || cam->y_res % threadDivisionsInY) return 1;
	string name = string(workdir) + "/ref_raster_frame";
	PPM_Image I(cam->x_res, cam->y_res);
	RasterDetails rd(*scene, *cam, name, I);
	Rasterizer r;
	std::streambuf *o = std::cout.rdbuf(nullptr);
	for (int i = 0; i < threadDivisionsInX; ++i)
		for (int j = 0; j < threadDivisionsInY; ++j) {
			if (mode == 0) r.Normal_Raster(rd, i, j);
			else if (mode == 1) r.Anti_Aliasing(rd, i, j);
			else r.Depth_Of_Field_Effect(rd, i, j);
		}
	std::cout.rdbuf(o);
	std::cout.clear();
	for (unsigned i = 0; i < cam->y_res; ++i)
		for (unsigned j = 0; j < cam->x_res; ++j) {
			const Pixel &p = I(i, j);
			unsigned char *q = pixels + 3 * ((size_t)i * cam->x_res + j);
			q[0] = p.r; q[1] = p.g; q[2] = p.b;
		}
	return 0;
}

// ToneMapRadiosity (rasterizer.cpp:90-109, file-static) on n colours -> n × 3 bytes
extern "C" void ref_tone_map_radiosity(long n, const double *rgb, unsigned char *out) {
	for (long k = 0; k < n; ++k) {
		Pixel p = ToneMapRadiosity(Color(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]));
		out[3 * k] = p.r; out[3 * k + 1] = p.g; out[3 * k + 2] = p.b;
	}
}
