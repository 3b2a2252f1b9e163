# Helpers for julia/test/amdgpu.jl -- the ROCArray counterparts of the reference's
# `run_cuda` / `cuda_cpu_agree` (/root/reference/test/util.jl:1-34).  UNEXECUTED SOURCE (no Julia
# runtime in this image); the same comparisons run for real, against the CPU oracle, in
# tests/test_parity_gpu.py::test_device_equals_oracle (Python harness over the same C ABI).
using Adapt, AMDGPU

"Move every array argument to the GPU, call `f`, return its (device) result."
on_device(f, args...) = f(adapt(ROCArray, args)...)

"`f` on ROCArrays agrees with `f` on CPU arrays, compared with Julia's `≈` (rtol = sqrt(eps))."
function roc_cpu_agree(f, args...)
    expected = f(args...)
    actual = on_device(f, args...)
    return agrees(actual, expected)
end

agrees(actual::AbstractArray, expected::AbstractArray) = Array(actual) ≈ expected

function agrees(actual::NamedTuple, expected::NamedTuple)
    host = adapt(Array, actual)
    for name in propertynames(expected)
        a, e = getproperty(host, name), getproperty(expected, name)
        if !(a ≈ e)
            @error "field differs between ROCArray and CPU result" name a e
            return false
        end
    end
    return true
end
