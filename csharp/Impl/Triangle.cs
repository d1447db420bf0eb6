// Triangle.cs — the mesh primitive BASELINE config 4 needs (source only, see HipRenderBackend.cs).  The reference has no triangle or mesh
// hittable (Impl/Hittables holds Sphere, Quad, Box, Capsule, Cylinder, Disk, InfinitePlane, XY/XZ/YZPlane, ConstantDensityMedium;
// Core/Todos.md:20-22 lists "Import 3D models" as a TODO), so this is NEW and defined by analogy with Quad (Quad.cs:37-151) so that
// materials behave the same on both: the same plane test with the same |n.d| <= 0.001 cull and the same t formula, the inside test
// u >= 0, v >= 0, u + v <= 1 on the coordinates of the inverted (E1, E2, N) basis, the geometric normal never flipped towards the ray,
// OutsideFace = n.d < 0, UV = (u, v), bounding volume of the three vertices padded by 1e-4.  oracle/oracle.cpp (Quad with triangle =
// true) and the kernels' try_quad<true> (rt_device.h) are the same arithmetic; rt_triangle in include/rt_hip.h is its record.
//
// Where it goes: Ararem.RayTracer.Impl/Hittables/Triangle.cs.
using System;
using System.Numerics;
using Ararem.RayTracer.Core;
using Ararem.RayTracer.Core.Acceleration;
using static System.Numerics.Vector3;

namespace Ararem.RayTracer.Impl.Hittables;

public sealed partial class Triangle : SingleMaterialHittable
{
	private readonly Matrix4x4 localToTriangleMatrix;

	public Triangle(Vector3 v0, Vector3 v1, Vector3 v2)
	{
		V0             = v0;
		E1             = v1 - v0;
		E2             = v2 - v0;
		Normal         = Normalize(Cross(E1, E2));
		BoundingVolume = AxisAlignedBoundingBox.Encompass(V0, V0 + E1, V0 + E2).WithPadding(.0001f);
		Matrix4x4 triangleToWorld = new(E1.X, E1.Y, E1.Z, 0, E2.X, E2.Y, E2.Z, 0, Normal.X, Normal.Y, Normal.Z, 0, 0, 0, 0, 1);
		if (!Matrix4x4.Invert(triangleToWorld, out localToTriangleMatrix)) throw new ArithmeticException("Could not invert Triangle to world transformation matrix (degenerate triangle)");
	}

	public Vector3 V0     { get; }
	public Vector3 E1     { get; }
	public Vector3 E2     { get; }
	public Vector3 Normal { get; }

	public override AxisAlignedBoundingBox BoundingVolume { get; }

	public override HitRecord? TryHit(Ray ray, float kMin, float kMax)
	{
		if (!Intersect(ray, kMin, kMax, out float t, out float u, out float v, out float normDotDir)) return null;
		Vector3 worldPoint = ray.PointAt(t);
		return new HitRecord(ray, worldPoint, worldPoint - V0, Normal, t, normDotDir < 0, new Vector2(u, v), this, Material);
	}

	public override bool FastTryHit(Ray ray, float kMin, float kMax) => Intersect(ray, kMin, kMax, out _, out _, out _, out _);

	private bool Intersect(Ray ray, float kMin, float kMax, out float t, out float u, out float v, out float normDotDir)
	{
		t          = u = v = 0;
		normDotDir = Dot(Normal, ray.Direction);
		if (MathF.Abs(normDotDir) <= 0.001f) return false;
		t = -Dot(Normal, ray.Origin - V0) / normDotDir;
		if ((t < kMin) || (t > kMax)) return false;
		Vector3 localPoint = ray.PointAt(t) - V0;
		u = (localPoint.X * localToTriangleMatrix.M11) + (localPoint.Y * localToTriangleMatrix.M21) + (localPoint.Z * localToTriangleMatrix.M31) + localToTriangleMatrix.M41;
		v = (localPoint.X * localToTriangleMatrix.M12) + (localPoint.Y * localToTriangleMatrix.M22) + (localPoint.Z * localToTriangleMatrix.M32) + localToTriangleMatrix.M42;
		return (u >= 0) && (v >= 0) && (u + v <= 1);      // NaN fails every comparison
	}
}
