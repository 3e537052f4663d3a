// TEST INFRASTRUCTURE — the handle ref_driver.cpp / ref_adapter.cpp pass to Python: the reference's own Scene,
// Camera and Radiosity objects (Render2/ray_tracer.h), nothing else.
#pragma once
#include <vector>
struct RefScene {
	Scene scene;
	Camera cam;
	Radiosity *rad = nullptr;
	std::vector<QuadTreeNode *> elems;
	bool bsp_built = false;
};
