// DeviceFlatteners.cs — IDeviceFlattenable for every builtin class of Ararem.RayTracer.Impl (source only, see HipRenderBackend.cs).
//
// Where it goes: Ararem.RayTracer.Impl/Backends/DeviceFlatteners.cs.  Each class below is the `partial` other half of the reference class
// of the same name (the reference declarations gain the `partial` modifier, nothing else): a flattener copies exactly the fields its
// TryHit / FastTryHit / Scatter / CalculateColour read, INCLUDING the values the constructor caches (localToQuadMatrix, negPointDotNormal,
// WorldToBoxTransform, uvMatrix, ...), so the device computes with the same numbers the CPU path would.  Field names on the right-hand
// sides are the reference's (file:line cited per class); record layouts are include/rt_hip.h's.
// CustomFunctionTexture, GreyscaleNoiseTexture over a Module other than Perlin, lights with a hand-written attenuation lambda and any
// user-defined subclass implement nothing here: DeviceSceneBuilder.IdOf throws NotSupportedException for them.
using System;
using System.Numerics;
using Ararem.RayTracer.Core;
using Ararem.RayTracer.Core.Backends;
using SharpNoise.Modules;

namespace Ararem.RayTracer.Impl.Hittables
{
	// SingleMaterialHittable.cs:8-22
	public abstract partial class SingleMaterialHittable : IDeviceMaterialOwner
	{
		Material IDeviceMaterialOwner.DeviceMaterial => Material;
	}

	// Sphere.cs:17-22
	public sealed unsafe partial class Sphere : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitSphere;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Sphere s = new() { Radius = Radius };
			RtHip.Put(s.Centre, Centre);
			b.Spheres.Add(s);
			return (uint)b.Spheres.Count - 1;
		}
	}

	// Quad.cs:37-55 — inv_col1 / inv_col2 are the two columns of localToQuadMatrix that TryHit reads (:107-108)
	public sealed unsafe partial class Quad : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitQuad;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Quad q = new();
			RtHip.Put(q.Origin, Origin); RtHip.Put(q.Normal, Normal);
			q.InvCol1[0] = localToQuadMatrix.M11; q.InvCol1[1] = localToQuadMatrix.M21; q.InvCol1[2] = localToQuadMatrix.M31; q.InvCol1[3] = localToQuadMatrix.M41;
			q.InvCol2[0] = localToQuadMatrix.M12; q.InvCol2[1] = localToQuadMatrix.M22; q.InvCol2[2] = localToQuadMatrix.M32; q.InvCol2[3] = localToQuadMatrix.M42;
			b.Quads.Add(q);
			return (uint)b.Quads.Count - 1;
		}
	}

	// Triangle.cs (new, csharp/Impl/Triangle.cs)
	public sealed unsafe partial class Triangle : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitTriangle;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Triangle t = new();
			RtHip.Put(t.V0, V0); RtHip.Put(t.E1, E1); RtHip.Put(t.E2, E2); RtHip.Put(t.Normal, Normal);
			t.InvCol1[0] = localToTriangleMatrix.M11; t.InvCol1[1] = localToTriangleMatrix.M21; t.InvCol1[2] = localToTriangleMatrix.M31; t.InvCol1[3] = localToTriangleMatrix.M41;
			t.InvCol2[0] = localToTriangleMatrix.M12; t.InvCol2[1] = localToTriangleMatrix.M22; t.InvCol2[2] = localToTriangleMatrix.M32; t.InvCol2[3] = localToTriangleMatrix.M42;
			b.Triangles.Add(t);
			return (uint)b.Triangles.Count - 1;
		}
	}

	// InfinitePlane.cs:12-20 (the normal is NOT normalised by the constructor; BoundingVolume is infinite, :28: the library keeps it outside the BVH)
	public sealed unsafe partial class InfinitePlane : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitInfinitePlane;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.InfinitePlane p = new() { NegPointDotNormal = negPointDotNormal };
			RtHip.Put(p.Point, Point); RtHip.Put(p.Normal, Normal);
			b.InfinitePlanes.Add(p);
			return (uint)b.InfinitePlanes.Count - 1;
		}
	}

	// XYPlane.cs:19-29  (a = x, b = y, c = z)
	public sealed partial class XYPlane : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitXyPlane;
		public uint Flatten(DeviceSceneBuilder b)
		{
			b.XyPlanes.Add(new RtHip.AxisPlane { ALo = XLow, AHi = XHigh, BLo = YLow, BHi = YHigh, C = Z });
			return (uint)b.XyPlanes.Count - 1;
		}
	}

	// XZPlane.cs:19-29  (a = x, b = z, c = y)
	public sealed partial class XZPlane : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitXzPlane;
		public uint Flatten(DeviceSceneBuilder b)
		{
			b.XzPlanes.Add(new RtHip.AxisPlane { ALo = XLow, AHi = XHigh, BLo = ZLow, BHi = ZHigh, C = Y });
			return (uint)b.XzPlanes.Count - 1;
		}
	}

	// YZPlane.cs:17-26  (a = y, b = z, c = x)
	public sealed partial class YZPlane : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitYzPlane;
		public uint Flatten(DeviceSceneBuilder b)
		{
			b.YzPlanes.Add(new RtHip.AxisPlane { ALo = YLow, AHi = YHigh, BLo = ZLow, BHi = ZHigh, C = X });
			return (uint)b.YzPlanes.Count - 1;
		}
	}

	// Disk.cs:23-41
	public sealed unsafe partial class Disk : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitDisk;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Disk d = new() { Radius = Radius, NegCentreDotNorm = negCentreDotNorm };
			RtHip.Put(d.Centre, Centre); RtHip.Put(d.Normal, Normal);
			d.InvCol1[0] = localToDiskMatrix.M11; d.InvCol1[1] = localToDiskMatrix.M21; d.InvCol1[2] = localToDiskMatrix.M31; d.InvCol1[3] = localToDiskMatrix.M41;
			d.InvCol2[0] = localToDiskMatrix.M12; d.InvCol2[1] = localToDiskMatrix.M22; d.InvCol2[2] = localToDiskMatrix.M32; d.InvCol2[3] = localToDiskMatrix.M42;
			b.Disks.Add(d);
			return (uint)b.Disks.Count - 1;
		}
	}

	// Box.cs:35-65 — world_to_box = WorldToBoxTransform rows 1..4, columns 1..3; box_to_world_rows = boxToWorldRow1..3 (:62-64)
	public sealed unsafe partial class Box : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitBox;
		public uint Flatten(DeviceSceneBuilder b)
		{
			Matrix4x4 w = WorldToBoxTransform;
			RtHip.Box x = new();
			x.WorldToBox[0] = w.M11; x.WorldToBox[1]  = w.M12; x.WorldToBox[2]  = w.M13;
			x.WorldToBox[3] = w.M21; x.WorldToBox[4]  = w.M22; x.WorldToBox[5]  = w.M23;
			x.WorldToBox[6] = w.M31; x.WorldToBox[7]  = w.M32; x.WorldToBox[8]  = w.M33;
			x.WorldToBox[9] = w.M41; x.WorldToBox[10] = w.M42; x.WorldToBox[11] = w.M43;
			RtHip.Put(x.BoxToWorldRows, boxToWorldRow1); RtHip.Put(x.BoxToWorldRows + 3, boxToWorldRow2); RtHip.Put(x.BoxToWorldRows + 6, boxToWorldRow3);
			b.Boxes.Add(x);
			return (uint)b.Boxes.Count - 1;
		}
	}

	// Capsule.cs:35-55 — uv_matrix = uvMatrix M11 M12 M13 M21 M22 M23 M31 M32 M33 (:48)
	public sealed unsafe partial class Capsule : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitCapsule;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Capsule c = new() { Radius = Radius };
			RtHip.Put(c.P1, P1); RtHip.Put(c.P2, P2);
			c.UvMatrix[0] = uvMatrix.M11; c.UvMatrix[1] = uvMatrix.M12; c.UvMatrix[2] = uvMatrix.M13;
			c.UvMatrix[3] = uvMatrix.M21; c.UvMatrix[4] = uvMatrix.M22; c.UvMatrix[5] = uvMatrix.M23;
			c.UvMatrix[6] = uvMatrix.M31; c.UvMatrix[7] = uvMatrix.M32; c.UvMatrix[8] = uvMatrix.M33;
			b.Capsules.Add(c);
			return (uint)b.Capsules.Count - 1;
		}
	}

	// Cylinder.cs:28-40
	public sealed unsafe partial class Cylinder : IDeviceHittable
	{
		public uint DeviceType => RtHip.HitCylinder;
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Cylinder c = new() { Radius = Radius };
			RtHip.Put(c.P1, P1); RtHip.Put(c.P2, P2);
			b.Cylinders.Add(c);
			return (uint)b.Cylinders.Count - 1;
		}
	}

	// ConstantDensityMedium.cs:23-30, :80 (BoundingVolume = Boundary's); its private VolumetricMaterial (:87-130) is flattened below
	public sealed partial class ConstantDensityMedium : IDeviceHittable, IDeviceMaterialOwner
	{
		public uint DeviceType => RtHip.HitMedium;
		Material IDeviceMaterialOwner.DeviceMaterial => InternalMaterial;
		public uint Flatten(DeviceSceneBuilder b)
		{
			(uint type, uint index) = b.Shape(Boundary);          // the boundary's own record; it need not be a scene object
			if (type == RtHip.HitMedium) throw new NotSupportedException("Cannot create a constant density volume using another volume");   // :25
			b.Media.Add(new RtHip.Medium { BoundaryType = type, BoundaryIndex = index, Density = Density, NegInvDensity = negInvDensity });
			return (uint)b.Media.Count - 1;
		}

		private sealed unsafe partial class VolumetricMaterial : IDeviceFlattenable
		{
			public uint Flatten(DeviceSceneBuilder b)
			{
				RtHip.Material m = new() { Type = RtHip.MatVolumetric };
				m.F[0] = Albedo.R; m.F[1] = Albedo.G; m.F[2] = Albedo.B; m.F[3] = Density;
				b.Materials.Add(m);
				return (uint)b.Materials.Count - 1;
			}
		}
	}
}

namespace Ararem.RayTracer.Impl.Textures
{
	// SolidColourTexture.cs:9-15
	public sealed unsafe partial class SolidColourTexture : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Texture t = new() { Type = RtHip.TexSolid };
			t.F[0] = Colour.R; t.F[1] = Colour.G; t.F[2] = Colour.B;
			b.Textures.Add(t);
			return (uint)b.Textures.Count - 1;
		}
	}

	// MarbleTexture.cs:27-54 (AccentColour / BaseColour are the fixed values of :36-39; the shared static Perlin module of :11-17 is what the kernels implement)
	public sealed unsafe partial class MarbleTexture : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Texture t = new() { Type = RtHip.TexMarble };
			t.F[0] = Scale; t.F[1] = NoiseScale; t.F[2] = NoiseStrength; t.F[3] = DropoffPower;
			b.Textures.Add(t);
			return (uint)b.Textures.Count - 1;
		}
	}

	// GreyscaleNoiseTexture.cs:11-20 — only over a SharpNoise Perlin module; any other Module is opaque code
	public sealed unsafe partial class GreyscaleNoiseTexture : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			if (Module is not Perlin p) throw new NotSupportedException($"GreyscaleNoiseTexture over {Module.GetType().FullName}: only a Perlin module has a device implementation");
			RtHip.Texture t = new() { Type = RtHip.TexGreyscalePerlin };
			t.F[0] = (float)p.Frequency; t.F[1] = (float)p.Lacunarity; t.F[2] = (float)p.Persistence; t.F[3] = p.OctaveCount; t.F[4] = p.Seed;
			b.Textures.Add(t);
			return (uint)b.Textures.Count - 1;
		}
	}
}

namespace Ararem.RayTracer.Impl.Materials
{
	// StandardMaterial.cs:16-61
	public sealed unsafe partial class StandardMaterial : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Material m = new() { Type = RtHip.MatStandard };
			m.Tex[0] = b.IdOf(Albedo); m.Tex[1] = b.IdOf(Emission); m.F[0] = Diffusion;
			b.Materials.Add(m);
			return (uint)b.Materials.Count - 1;
		}
	}

	// RefractiveMaterial.cs:20-34; EmissiveRefractiveMaterial.cs:30-40 derives from it and overrides
	public unsafe partial class RefractiveMaterial : IDeviceFlattenable
	{
		public virtual uint Flatten(DeviceSceneBuilder b)
		{
			if (GetType() != typeof(RefractiveMaterial)) throw new NotSupportedException($"{GetType().FullName}: user-defined Materials are rejected by the HIP backend");
			RtHip.Material m = new() { Type = RtHip.MatRefractive, Flags = AlternateRefractionMode ? RtHip.MatFlagAlternateRefraction : 0u };
			m.Tex[0] = b.IdOf(Tint); m.F[0] = RefractiveIndex;
			b.Materials.Add(m);
			return (uint)b.Materials.Count - 1;
		}
	}

	public sealed unsafe partial class EmissiveRefractiveMaterial
	{
		public override uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Material m = new()
			{
					Type  = RtHip.MatEmissiveRefractive,
					Flags = (AlternateRefractionMode ? RtHip.MatFlagAlternateRefraction : 0u) | (DirectEmission ? RtHip.MatFlagDirectEmission : 0u)
			};
			m.Tex[0] = b.IdOf(Tint); m.Tex[1] = b.IdOf(Emission); m.F[0] = RefractiveIndex;
			b.Materials.Add(m);
			return (uint)b.Materials.Count - 1;
		}
	}

	// PhongMaterial.cs:11-23 (not sealed in the reference: a subclass would change Scatter/CalculateColour, so it is rejected)
	public unsafe partial class PhongMaterial : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			if (GetType() != typeof(PhongMaterial)) throw new NotSupportedException($"{GetType().FullName}: user-defined Materials are rejected by the HIP backend");
			RtHip.Material m = new() { Type = RtHip.MatPhong };
			m.F[0] = SpecularColour.R; m.F[1] = SpecularColour.G; m.F[2] = SpecularColour.B;
			m.F[3] = DiffuseColour.R;  m.F[4] = DiffuseColour.G;  m.F[5] = DiffuseColour.B;
			m.F[6] = AmbientColour.R;  m.F[7] = AmbientColour.G;  m.F[8] = AmbientColour.B;
			m.F[9] = Shininess;
			b.Materials.Add(m);
			return (uint)b.Materials.Count - 1;
		}
	}
}

namespace Ararem.RayTracer.Impl.Lights
{
	// SimpleLightBase.cs:16-34, :132.  DistanceAttenuationFunc is a delegate (:139); the factories (:83-129) now return a DistanceAttenuation record
	// (csharp/Impl/DistanceAttenuation.cs) that converts implicitly to the delegate, so existing initialisers compile unchanged and the
	// flattener reads Kind / P0..P2 instead of probing a lambda.  A hand-written lambda has no record: rejected.
	public abstract unsafe partial class SimpleLightBase
	{
		/// <summary>The record behind DistanceAttenuationFunc when it came from one of the factories; null for a hand-written delegate.</summary>
		public DistanceAttenuation? Attenuation { get; init; } = DistanceAttenuation.ExponentialDecay(5);

		private protected RtHip.Light FlattenCommon(uint type)
		{
			if (Attenuation is not { } a || !ReferenceEquals(a.Func, DistanceAttenuationFunc))
				throw new NotSupportedException("DistanceAttenuationFunc is an arbitrary delegate; only the SimpleLightBase factory families (and a constant) can run on the device");
			RtHip.Light l = new() { Type = type, AttenuationRadius = AttenuationRadius, CutoffRadius = CutoffRadius, AttenKind = a.Kind };
			RtHip.Put(l.Position, Position); RtHip.Put(l.Colour, Colour);
			l.AttenP[0] = a.P0; l.AttenP[1] = a.P1; l.AttenP[2] = a.P2;
			return l;
		}
	}

	// PointLight.cs:6-10
	public sealed partial class PointLight : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b) { b.Lights.Add(FlattenCommon(RtHip.LightPoint)); return (uint)b.Lights.Count - 1; }
	}

	// DiffuseSphereLight.cs:10-13 (the live `case 4` of GetShadowRayForHit, :88-142, is what the kernels implement)
	public partial class DiffuseSphereLight : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			if (GetType() != typeof(DiffuseSphereLight)) throw new NotSupportedException($"{GetType().FullName}: user-defined Lights are rejected by the HIP backend");
			RtHip.Light l = FlattenCommon(RtHip.LightDiffuseSphere);
			l.DiffusionRadius = DiffusionRadius;
			b.Lights.Add(l);
			return (uint)b.Lights.Count - 1;
		}
	}

	// DiffuseShapedLight.cs:9-15, GetShadowRayForHit :69-105 (rejection sampling inside Shape.BoundingVolume)
	public sealed unsafe partial class DiffuseShapedLight : IDeviceFlattenable
	{
		public uint Flatten(DeviceSceneBuilder b)
		{
			RtHip.Light l = FlattenCommon(RtHip.LightDiffuseShaped);
			(l.ShapeType, l.ShapeIndex) = b.Shape(Shape);
			if (l.ShapeType == RtHip.HitMedium) throw new NotSupportedException("DiffuseShapedLight over a ConstantDensityMedium is not supported on the device");
			RtHip.Put(l.ShapeAabbMin, Shape.BoundingVolume.Min); RtHip.Put(l.ShapeAabbMax, Shape.BoundingVolume.Max);
			b.Lights.Add(l);
			return (uint)b.Lights.Count - 1;
		}
	}
}

namespace Ararem.RayTracer.Impl.Skyboxes
{
	// DefaultSkyBox.cs:8-16
	public sealed partial class DefaultSkyBox : IDeviceSkyBox
	{
		public RtHip.SkyBox FlattenSkyBox() => new() { Type = RtHip.SkyDefault };
	}

	// SingleColourSkyBox.cs:8-21 (not sealed in the reference: a subclass is rejected)
	public unsafe partial class SingleColourSkyBox : IDeviceSkyBox
	{
		public RtHip.SkyBox FlattenSkyBox()
		{
			if (GetType() != typeof(SingleColourSkyBox)) throw new NotSupportedException($"{GetType().FullName}: user-defined SkyBoxes are rejected by the HIP backend");
			RtHip.SkyBox s = new() { Type = RtHip.SkySingleColour };
			RtHip.Put(s.Colour, Colour);
			return s;
		}
	}
}
