// DistanceAttenuation.cs — the tagged record the attenuation factories of SimpleLightBase (SimpleLightBase.cs:83-132) return in place of a
// bare lambda (source only, see HipRenderBackend.cs).  It converts implicitly to SimpleLightBase.DistanceAttenuationDelegate (:139), so
// `DistanceAttenuationFunc = SimpleLightBase.LinearDistanceAttenuation()` keeps compiling and the CPU path keeps calling Func; the HIP
// backend reads Kind / P0..P2 (rt_attenuation_kind, include/rt_hip.h) instead of probing a delegate.
//
// Where it goes: Ararem.RayTracer.Impl/Lights/DistanceAttenuation.cs, and the factory bodies at SimpleLightBase.cs:83-129 become
//     public static DistanceAttenuation LinearDistanceAttenuation()                                   => DistanceAttenuation.Linear();
//     public static DistanceAttenuation PowerDistanceAttenuation(float power, bool stayHigh = false)  => DistanceAttenuation.Power(power, stayHigh);
//     public static DistanceAttenuation LogisticsCurveDistanceAttenuation(float midpoint = .5f, float steepness = 16f) => DistanceAttenuation.Logistic(midpoint, steepness);
//     public static DistanceAttenuation LogisticsCurveDistanceAttenuation(float l, float k, float x0) => DistanceAttenuation.LogisticLkx(l, k, x0);
//     public static DistanceAttenuation ExponentialDecayDistanceAttenuation(float a)                  => DistanceAttenuation.ExponentialDecay(a);
// A light initialised with a record sets both DistanceAttenuationFunc (= record.Func) and Attenuation (= record) through the init accessor of
// DistanceAttenuationFunc; a light initialised with a hand-written lambda leaves Attenuation unmatched and is rejected by the flattener.
using System;
using static System.MathF;

namespace Ararem.RayTracer.Impl.Lights;

public sealed record DistanceAttenuation(uint Kind, float P0, float P1, float P2, SimpleLightBase.DistanceAttenuationDelegate Func)
{
	public static implicit operator SimpleLightBase.DistanceAttenuationDelegate(DistanceAttenuation a) => a.Func;

	/// <summary>RT_ATTEN_CONSTANT: e.g. the Demo scene's `(_, _) => 1f` (BuiltinScenes.cs:146), written as DistanceAttenuation.Constant(1f)</summary>
	public static DistanceAttenuation Constant(float value) => new(0, value, 0, 0, (_, _) => value);

	/// <summary>RT_ATTEN_LINEAR, SimpleLightBase.cs:83</summary>
	public static DistanceAttenuation Linear() => new(1, 0, 0, 0, static (_, normDist) => 1 - normDist);

	/// <summary>RT_ATTEN_POWER / RT_ATTEN_POWER_STAY_HIGH, SimpleLightBase.cs:95-103</summary>
	public static DistanceAttenuation Power(float power, bool stayHighInitially = false) => stayHighInitially
			? new DistanceAttenuation(3, power, 0, 0, (_, normDist) => 1 - Pow(normDist, power))
			: new DistanceAttenuation(2, power, 0, 0, (_, normDist) => Pow(1 - normDist, power));

	/// <summary>RT_ATTEN_LOGISTIC, SimpleLightBase.cs:115</summary>
	public static DistanceAttenuation Logistic(float midpoint = .5f, float steepness = 16f) => new(4, midpoint, steepness, 0, (_, normDist) => 1f / (1 + Pow(E, steepness * (normDist - midpoint))));

	/// <summary>RT_ATTEN_LOGISTIC_LKX, SimpleLightBase.cs:125</summary>
	public static DistanceAttenuation LogisticLkx(float l, float k, float x0) => new(5, l, k, x0, (_, normDist) => l / (1f + Pow(E, -k * (normDist - x0))));

	/// <summary>RT_ATTEN_EXP_DECAY, SimpleLightBase.cs:129 (the default, a = 5, :132)</summary>
	public static DistanceAttenuation ExponentialDecay(float a) => new(6, a, 0, 0, (_, normDist) => Pow(E, a * -normDist));
}
