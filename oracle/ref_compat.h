// TEST INFRASTRUCTURE — not part of the product.
//
// Force-included (-include) in front of every reference translation unit when
// oracle/Makefile compiles the reference's own sources *in place* from
// /root/reference/Render2 with ROCm clang++.  The reference is an MSVC program;
// MSVC lets an rvalue bind to a non-const lvalue reference, GCC/Clang do not.
// Three helper functions of the reference (display/meshing path, never executed
// by the hot path) are called that way:
//   doesPointLieOnLine / isPointNearLine  radiosity_helper.h:95,104 (calls :175-220)
//   sideOfLine3D                          ray.h:66 (calls radiosity_helper.cpp:86-128)
// Declaring const-reference overloads ahead of the headers makes those call
// sites resolve; the overloads (defined in ref_driver.cpp) copy their arguments
// into lvalues and forward to the reference's own functions.  Nothing of the
// reference is replaced, copied or edited.
#pragma once
struct Vec3;
bool doesPointLieOnLine(const Vec3 &, const Vec3 &, const Vec3 &);
bool isPointNearLine(const Vec3 &, const Vec3 &, const Vec3 &);
int sideOfLine3D(const Vec3 &, const Vec3 &);
