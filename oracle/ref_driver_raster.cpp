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
