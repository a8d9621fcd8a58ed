# MI355DSP.jl -- thin Julia host for libmi355dsp.so (the C ABI of include/mi355dsp.h).
#
# STATUS: written against the header, NOT executed -- the build image has no Julia.  The same C ABI is exercised
# by the Python/ctypes host (dsp.jl_amd/) and its GPU parity tests; this file is the `ccall` twin a DSP.jl
# maintainer would use.  tests/test_abi_cpu.py checks every `ccall` here against the ctypes prototype of the same
# entry point: symbol exported, argument count, and every argument TYPE (Cint / Int64 / Cdouble / Csize_t / pointer kind).
#
# It keeps DSP.jl's names, argument order, defaults, promotion rules, result types (`Periodogram`, `Spectrogram` with
# `power` / `freq` / `time`) and exception types for the hot path -- filt / filt! / fftfilt / fftfilt! / tdfilt / tdfilt! /
# conv / conv! / xcorr / hilbert / periodogram / welch_pgram / welch_pgram! / spectrogram / stft / arraysplit / FIRFilter /
# resample / DF2TFilter / filtfilt / the mt_* family -- and does nothing else: no CUDA.jl / AMDGPU.jl array dispatch, every
# call goes straight to a hand-written HIP kernel through `ccall`.
#
# Host `Array`s go through the library's host-pointer pipelines (mdsp_{ols,welch,stft,fir}_exec_host: H2D || kernel || D2H on
# three streams; `pin!` page-locks an Array so the DMA engines use it directly); `DeviceArray` keeps data resident in HBM
# between calls.  The function-style calls take their plans from the library's own LRU (mdsp_*_plan_cached, one list per
# calling thread); the multi-GPU part is `Comm` (RCCL behind the C ABI).  Reference locations are DSP.jl v0.8.5 `src/`.
module MI355DSP

using LinearAlgebra: SymTridiagonal, eigen

export DeviceArray, upload, download, filt, filt!, fftfilt, fftfilt!, tdfilt, tdfilt!, conv, conv!, xcorr, hilbert,
       Periodogram, Spectrogram, power, freq, arraysplit, periodogram, WelchConfig, welch_pgram, welch_pgram!,
       spectrogram, stft, FIRFilter, resample, resample_filter, reset!, setphase!, timedelay, inputlength, outputlength,
       DF2TFilter, filtfilt, nextfastfft, optimalfftfiltlength, Comm, welch_channel_mean, welch_reset!, welch_accumulate!,
       welch_finalize, welch_allreduce!, pin!, unpin!, MTConfig, mt_pgram, mt_pgram!, MTSpectrogramConfig, mt_spectrogram,
       mt_spectrogram!, MTCrossSpectraConfig, mt_cross_power_spectra, mt_cross_power_spectra!, MTCoherenceConfig,
       mt_coherence, mt_coherence!, CrossPowerSpectra, Coherence, coherence, dpss

const lib = get(ENV, "MI355DSP_LIB", joinpath(@__DIR__, "..", "dsp.jl_amd", "libmi355dsp.so"))

# ---------------------------------------------------------------------------------------------- status -> exception
const MDSP_F32, MDSP_F64, MDSP_C32, MDSP_C64 = Cint(0), Cint(1), Cint(2), Cint(3)
const ENGINE_AUTO, ENGINE_FUSED, ENGINE_ROCFFT = Cint(0), Cint(1), Cint(2)
const OLS_FILT, OLS_CONV = Cint(0), Cint(1)
const HOST_PINNED = Cint(1)

struct UnsupportedError <: Exception
    msg::String
end
struct DeviceError <: Exception
    msg::String
end

lasterr() = unsafe_string(ccall((:mdsp_last_error_string, lib), Cstring, ()))

function check(status::Cint)
    status == 0 && return nothing
    msg = lasterr()
    status == -1 && throw(ArgumentError(msg))          # dspbase.jl:28-33, filt.jl:474,531, periodograms.jl:396,564,876
    status == -2 && throw(DomainError(nothing, msg))   # periodograms.jl:44-45,397,565; stream_filt.jl:194,217
    status == -3 && throw(DimensionMismatch(msg))      # periodograms.jl:255,735-737
    status == -4 && throw(AssertionError(msg))         # stream_filt.jl:634,718,722
    status == -5 && throw(UnsupportedError(msg))       # caller should fall back to DSP.jl itself
    status == -7 && throw(OutOfMemoryError())
    throw(DeviceError(msg))
end

mdtype(::Type{Float32}) = MDSP_F32
mdtype(::Type{Float64}) = MDSP_F64
mdtype(::Type{ComplexF32}) = MDSP_C32
mdtype(::Type{ComplexF64}) = MDSP_C64
const JLTYPE = (Float32, Float64, ComplexF32, ComplexF64)

# element-type rules, util.jl:92-104
fftintype(::Type{T}) where {T<:Union{Float32,Float64,ComplexF32,ComplexF64}} = T
fftintype(::Type{T}) where {T<:Real} = Float64
fftintype(::Type{T}) where {T<:Complex} = ComplexF64
fftouttype(::Type{T}) where {T<:Union{ComplexF32,ComplexF64}} = T
fftouttype(::Type{Float32}) = ComplexF32
fftouttype(::Type{T}) where {T<:Union{Real,Complex}} = ComplexF64
fftabs2type(::Type{T}) where {T<:Union{Float32,ComplexF32}} = Float32
fftabs2type(::Type{T}) where {T<:Union{Real,Complex}} = Float64

# ---------------------------------------------------------------------------------------------- library state / diagnostics
init(device::Integer=0) = check(ccall((:mdsp_init, lib), Cint, (Cint,), device))
shutdown() = check(ccall((:mdsp_shutdown, lib), Cint, ()))                         # borrowed plans and staging buffers die here
version() = Int(ccall((:mdsp_version, lib), Cint, ()))
function device_count()
    n = Ref{Cint}(0)
    check(ccall((:mdsp_device_count, lib), Cint, (Ref{Cint},), n)); Int(n[])
end
reload_tunables() = check(ccall((:mdsp_reload_tunables, lib), Cint, ()))           # tuning tools only: re-read the fifteen MDSP_* variables from ENV
set_knob(name::AbstractString, value::Integer) = check(ccall((:mdsp_set_knob, lib), Cint, (Cstring, Cint, Cint), name, value, 0))   # experiments: not environment variables
unset_knob(name::AbstractString) = check(ccall((:mdsp_set_knob, lib), Cint, (Cstring, Cint, Cint), name, 0, 1))
debug_knobs() = ccall((:mdsp_debug_knobs, lib), Cint, ()) != 0
synchronize(stream::Ptr{Cvoid}=C_NULL) = check(ccall((:mdsp_stream_synchronize, lib), Cint, (Ptr{Cvoid},), stream))
function plan_cache_stats()
    e, h, m = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:mdsp_plan_cache_stats, lib), Cint, (Ref{Int64}, Ref{Int64}, Ref{Int64}), e, h, m))
    (entries = Int(e[]), hits = Int(h[]), misses = Int(m[]))
end
plan_cache_clear() = check(ccall((:mdsp_plan_cache_clear, lib), Cint, ()))
host_pipeline_trim() = check(ccall((:mdsp_host_pipeline_trim, lib), Cint, ()))   # frees the host-array pipelines' device / pinned buffers
function plan_cache_partitions()
    p, r = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:mdsp_plan_cache_partitions, lib), Cint, (Ref{Int64}, Ref{Int64}), p, r))
    (partitions = Int(p[]), reaped = Int(r[]))
end
# Julia tasks migrate between OS threads, the library's binding is thread-local: `with_plan_context` pins the task to its thread for the duration
# (sticky, restored afterwards), so the binding is made and restored on the SAME thread.  Calls nest: the task keeps a stack of its contexts in
# task-local storage, leaving a level re-binds the level above (not 0), and a context derived from the task (the default id) is released -- its
# plans freed with it -- only when the OUTERMOST level that uses it returns (a nested default-id call derives the same id: releasing it there would
# free plans the outer level still holds).  Pass an id of your own (and call plan_cache_release_context when its work is done) to keep plans across
# calls.  The binding belongs to the OS thread: `f` must not yield to another task that uses the library on this thread without a context of its
# own -- that task would run under this binding while `f` is suspended (wrap it in with_plan_context too: it re-binds on entry and on exit).
plan_cache_set_context(id::Integer) = check(ccall((:mdsp_plan_cache_set_context, lib), Cint, (UInt64,), UInt64(id)))
plan_cache_release_context(id::Integer) = check(ccall((:mdsp_plan_cache_release_context, lib), Cint, (UInt64,), UInt64(id)))
const _CTX_KEY = :mi355dsp_plan_contexts
function with_plan_context(f, id::Union{Integer,Nothing}=nothing)
    t = current_task()
    own = id === nothing
    ctx = own ? (objectid(t) % UInt64) | (UInt64(1) << 62) : UInt64(id)
    stack = get!(() -> UInt64[], task_local_storage(), _CTX_KEY)::Vector{UInt64}
    was_sticky = t.sticky
    t.sticky = true                  # no migration between the bindings below
    push!(stack, ctx)
    plan_cache_set_context(ctx)
    try
        return f()
    finally
        pop!(stack)
        plan_cache_set_context(isempty(stack) ? UInt64(0) : last(stack))   # back to the enclosing level's binding
        own && !(ctx in stack) && plan_cache_release_context(ctx)          # outermost use of the derived context only
        t.sticky = was_sticky
    end
end

# HIP events on the library's launch stream (what bench.py times with)
mutable struct Event
    h::Ptr{Cvoid}
    function Event()
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mdsp_event_create, lib), Cint, (Ref{Ptr{Cvoid}},), p))
        e = new(p[])
        finalizer(x -> ccall((:mdsp_event_destroy, lib), Cint, (Ptr{Cvoid},), x.h), e)
        e
    end
end
record!(e::Event, stream::Ptr{Cvoid}=C_NULL) = (check(ccall((:mdsp_event_record, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), e.h, stream)); e)
function elapsed_ms(a::Event, b::Event)
    ms = Ref{Cfloat}(0)
    check(ccall((:mdsp_event_elapsed_ms, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Cfloat}), a.h, b.h, ms)); Float64(ms[])
end

# ---------------------------------------------------------------------------------------------- device memory
mutable struct DeviceArray{T,N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N,Int}
    function DeviceArray{T}(dims::NTuple{N,Int}) where {T,N}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mdsp_malloc, lib), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, prod(dims) * sizeof(T)))
        a = new{T,N}(p[], dims)
        finalizer(x -> ccall((:mdsp_free, lib), Cint, (Ptr{Cvoid},), x.ptr), a)
        a
    end
end
DeviceArray{T}(dims::Int...) where {T} = DeviceArray{T}(dims)
Base.size(a::DeviceArray) = a.dims
Base.size(a::DeviceArray, d::Integer) = d <= length(a.dims) ? a.dims[d] : 1
Base.length(a::DeviceArray) = prod(a.dims)
Base.eltype(::DeviceArray{T}) where {T} = T
Base.ndims(::DeviceArray{T,N}) where {T,N} = N
function Base.fill!(a::DeviceArray, byte::Integer)          # byte fill (zero the array with 0)
    check(ccall((:mdsp_memset, lib), Cint, (Ptr{Cvoid}, Cint, Csize_t, Ptr{Cvoid}), a.ptr, byte, length(a) * sizeof(eltype(a)), C_NULL)); a
end

function upload(x::Array{T,N}) where {T,N}
    d = DeviceArray{T}(size(x))
    GC.@preserve x check(ccall((:mdsp_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), d.ptr, pointer(x), sizeof(x), C_NULL))
    d
end
function download(d::DeviceArray{T,N}) where {T,N}
    x = Array{T,N}(undef, d.dims)
    GC.@preserve x check(ccall((:mdsp_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), pointer(x), d.ptr, sizeof(x), C_NULL))
    x
end
todevice(x::DeviceArray, ::Type{T}) where {T} = eltype(x) == T ? x : upload(convert(Array{T}, download(x)))
todevice(x::AbstractArray, ::Type{T}) where {T} = upload(convert(Array{T}, x))
back(y::DeviceArray, like::DeviceArray) = y
back(y::DeviceArray, like) = download(y)
ncolumns(x) = length(x) ÷ max(size(x, 1), 1)

# page-locked host memory: hipHostMalloc'ed arrays, or an existing Array page-locked in place (hipHostRegister)
function pinned_array(::Type{T}, dims::Int...) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mdsp_host_alloc, lib), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, prod(dims) * sizeof(T)))
    a = unsafe_wrap(Array, Ptr{T}(p[]), dims)
    finalizer(x -> ccall((:mdsp_host_free, lib), Cint, (Ptr{Cvoid},), pointer(x)), a)
    a
end
pin!(x::Array) = (check(ccall((:mdsp_host_register, lib), Cint, (Ptr{Cvoid}, Csize_t), pointer(x), sizeof(x))); x)
unpin!(x::Array) = (check(ccall((:mdsp_host_unregister, lib), Cint, (Ptr{Cvoid},), pointer(x))); x)

# ---------------------------------------------------------------------------------------------- index arithmetic
nextfastfft(n::Integer) = Int(ccall((:mdsp_nextfastfft, lib), Int64, (Int64,), n))                       # util.jl:134
optimalfftfiltlength(nb, nx) = Int(ccall((:mdsp_optimal_fft_len, lib), Int64, (Int64, Int64), nb, nx))  # dspbase.jl:268
framecount(len, n, noverlap) = Int(ccall((:mdsp_frame_count, lib), Int64, (Int64, Int64, Int64), len, n, noverlap))
outputlength(inlen::Integer, ratio::Union{Integer,Rational}, ϕ::Integer) =                                 # stream_filt.jl:317
    Int(ccall((:mdsp_outputlength, lib), Int64, (Int64, Int64, Int64, Int64), inlen, numerator(ratio), denominator(ratio), ϕ))
inputlength(outlen::Integer, ratio::Union{Integer,Rational}, ϕ::Integer, r::RoundingMode=RoundDown) =      # stream_filt.jl:358
    Int(ccall((:mdsp_inputlength, lib), Int64, (Int64, Int64, Int64, Int64, Cint), outlen, numerator(ratio), denominator(ratio), ϕ,
              (r == RoundUp || r == RoundFromZero) ? 1 : 0))
# block geometry of _fftfilt! (Filters/filt.jl:490, :504-517) for block `iblock` (0-based): what the parity tests compare bit for bit
function ols_block_geometry(nb, nfft, nx, iblock)
    o = ntuple(_ -> Ref{Int64}(0), 5)
    check(ccall((:mdsp_ols_block_geometry, lib), Cint, (Int64, Int64, Int64, Int64, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}),
                nb, nfft, nx, iblock, o[1], o[2], o[3], o[4], o[5]))
    (off = Int(o[1][]), npadbefore = Int(o[2][]), xstart = Int(o[3][]), n = Int(o[4][]), nout = Int(o[5][]))
end

const SMALL_FILT_CUTOFF = 66   # dspbase.jl:3

# ---------------------------------------------------------------------------------------------- overlap-save filt / conv
mutable struct OlsPlan
    h::Ptr{Cvoid}
    function OlsPlan(taps::Vector{T}, nfft::Integer, nx::Integer, mode::Integer, engine=ENGINE_AUTO) where {T}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve taps check(ccall((:mdsp_ols_plan_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Cint),
                                      p, pointer(taps), length(taps), nfft, nx, mdtype(T), mode, engine))
        o = new(p[])
        finalizer(x -> ccall((:mdsp_ols_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.h), o)
        o
    end
end
# (nfft, block length L, engine) as the reference would use them / as the fused engine actually executes a long filter
function plan_info(h::Ptr{Cvoid})
    nfft, L, e = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0)
    check(ccall((:mdsp_ols_plan_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Cint}), h, nfft, L, e))
    (nfft = Int(nfft[]), L = Int(L[]), engine = Int(e[]))
end
function plan_geometry(h::Ptr{Cvoid})
    nfft, L, parts = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0)
    check(ccall((:mdsp_ols_plan_geometry, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Cint}), h, nfft, L, parts))
    (nfft = Int(nfft[]), L = Int(L[]), partitions = Int(parts[]))
end
# the same without a plan or a device: what a plan for these arguments would execute (rows > 0: the rows form of the multi-pass engine)
function ols_geometry_for(nb::Integer, nfft::Integer, nx::Integer, ::Type{T}, mode::Integer=OLS_FILT, engine::Integer=ENGINE_AUTO) where {T}
    en, L, parts, eng, rows = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    check(ccall((:mdsp_ols_geometry_for, lib), Cint, (Int64, Int64, Int64, Cint, Cint, Cint, Ref{Int64}, Ref{Int64}, Ref{Cint}, Ref{Cint}, Ref{Cint}),
                nb, nfft, nx, mdtype(T), mode, engine, en, L, parts, eng, rows))
    (nfft = Int(en[]), L = Int(L[]), partitions = Int(parts[]), engine = Int(eng[]), rows = Int(rows[]))
end

# The function-style entry points build a plan per call in the reference (cheap FFTW plans); here the plan comes from the LIBRARY's LRU
# (mdsp_ols_plan_cached: keyed by device, thread, stream and the contents of the taps) -- ~45 us per call instead of ~1 ms.  The handle is
# borrowed: no finalizer, never destroyed from here.
function cached_ols_plan(taps::Vector{T}, nfft::Integer, nx::Integer, mode::Integer, engine=ENGINE_AUTO) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve taps check(ccall((:mdsp_ols_plan_cached, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Cint, Ptr{Cvoid}),
                                  p, pointer(taps), length(taps), nfft, nx, mdtype(T), mode, engine, C_NULL))
    p[]
end

function ols_exec!(y::DeviceArray, plan::Ptr{Cvoid}, x::DeviceArray, nout::Integer)
    nx = size(x, 1)
    check(ccall((:mdsp_ols_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                plan, x.ptr, nx, ncolumns(x), nx, y.ptr, nout, nout, C_NULL))
    y
end
function ols_exec_host!(y::Array, plan::Ptr{Cvoid}, x::Array, nout::Integer; pinned::Bool=false)
    nx = size(x, 1)
    GC.@preserve x y check(ccall((:mdsp_ols_exec_host, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Cint),
                                 plan, pointer(x), nx, ncolumns(x), nx, pointer(y), nout, nout, pinned ? HOST_PINNED : Cint(0)))
    y
end
# blocks [first, first + count) of the same block grid from a slice of the signal (time-axis split over GPUs; no collective)
function ols_exec_range!(ys::DeviceArray, plan::Ptr{Cvoid}, xs::DeviceArray, xs_first::Integer, nx::Integer, first::Integer, count::Integer, nout::Integer)
    check(ccall((:mdsp_ols_exec_range, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}),
                plan, xs.ptr, xs_first, length(xs), nx, ys.ptr, first, count, nout, C_NULL))
    ys
end
# the exact contents of `tmp1` before the forward transform (filt.jl:509-510) for blocks [first, first + count): parity tests
function ols_segment(plan::Ptr{Cvoid}, x::DeviceArray{T}, first::Integer, count::Integer) where {T}
    out = DeviceArray{T}((plan_info(plan).nfft, Int(count)))
    check(ccall((:mdsp_ols_segment, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                plan, x.ptr, size(x, 1), first, count, out.ptr, C_NULL))
    out
end

# fftfilt(b, x[, nfft]) / fftfilt!(out, b, x[, nfft])   Filters/filt.jl:458-476, _fftfilt! :479-521
# Device arrays: one launch sequence on resident data.  Host Arrays: the chunked H2D || kernel || D2H pipeline (same block grid, so
# both return bit-identical results).
# (filters of any length: beyond the partitioned kernels -- 16384 Float32 / 8192 Float64 taps -- the plan runs blocks of 2^20 points on the multi-pass engine)
function fftfilt(b::AbstractVector{H}, x::DeviceArray{T}, nfft::Integer=optimalfftfiltlength(length(b), length(x))) where {H<:Real,T<:Real}
    W = fftintype(promote_type(H, T))
    xd = todevice(x, W)
    nx = size(xd, 1)
    plan = cached_ols_plan(convert(Vector{W}, b), nfft, nx, OLS_FILT)
    ols_exec!(DeviceArray{W}(size(xd)), plan, xd, nx)
end
function fftfilt(b::AbstractVector{H}, x::AbstractArray{T}, nfft::Integer=optimalfftfiltlength(length(b), length(x));
                 pinned::Bool=false) where {H<:Real,T<:Real}
    W = fftintype(promote_type(H, T))
    xh = convert(Array{W}, x)                                  # column-major (n, cols...): exactly the layout the C ABI takes
    plan = cached_ols_plan(convert(Vector{W}, b), nfft, size(xh, 1), OLS_FILT)
    ols_exec_host!(similar(xh), plan, xh, size(xh, 1); pinned)
end
function fftfilt!(out::AbstractArray, b::AbstractVector{<:Real}, x::AbstractArray{<:Real}, nfft::Integer=optimalfftfiltlength(length(b), length(x)))
    size(out) == size(x) || throw(ArgumentError("out and x must be the same size"))            # filt.jl:474
    copyto!(out, fftfilt(b, x, nfft))
end

# filt(b, a::Number, x) / filt!(out, b, a, x) / tdfilt(h, x) / tdfilt!(out, h, x)   dspbase.jl:14-66, filt.jl:431-443 (FIR only on the device)
function filt(b::AbstractVector, a::Number, x::Union{AbstractArray{T},DeviceArray{T}}) where {T}
    isempty(b) && throw(ArgumentError("filter vector b must be non-empty"))
    a == 0 && throw(ArgumentError("filter vector a[1] must be nonzero"))
    W = fftintype(promote_type(eltype(b), typeof(a), T))
    taps = convert(Vector{real(W)}, a == 1 ? b : b ./ a)
    xd = todevice(x, W)
    y = DeviceArray{W}(size(xd))
    nx = size(xd, 1)
    GC.@preserve taps check(ccall((:mdsp_tdfir_exec, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                                  pointer(taps), length(taps), mdtype(W), xd.ptr, nx, ncolumns(xd), nx, y.ptr, nx, C_NULL))
    back(y, x)
end
filt(b::AbstractVector, a::AbstractVector, x) = length(a) == 1 ? filt(b, a[1], x) :
    throw(UnsupportedError("IIR filt(b, a, x) is a serial recursion; only FIR (scalar a) runs on the device"))
function filt!(out::AbstractArray, b::AbstractVector, a::Union{Number,AbstractVector}, x::AbstractArray)
    size(out) == size(x) || throw(ArgumentError("output size $(size(out)) must match input size $(size(x))"))   # dspbase.jl:32
    copyto!(out, filt(b, a, x))
end
tdfilt(h::AbstractVector{H}, x) where {H} = filt(h, one(H), x)                      # filt.jl:431
function tdfilt!(out::AbstractArray, h::AbstractVector, x::AbstractArray)
    size(out) == size(x) || throw(ArgumentError("out must be the same size as x"))
    copyto!(out, tdfilt(h, x))
end

# filt(b, x) / filt!(out, b, x): FFT path for real taps longer than SMALL_FILT_CUTOFF, time domain otherwise   filt.jl:445-446, :525-555
function filt(b::AbstractVector{<:Real}, x::Union{AbstractArray{<:Real},DeviceArray{<:Real}})
    length(b) > SMALL_FILT_CUTOFF ? fftfilt(b, x, optimalfftfiltlength(length(b), size(x, 1))) : tdfilt(b, x)
end
function filt!(out::AbstractArray, b::AbstractVector{<:Real}, x::AbstractArray{<:Real})
    size(out) == size(x) || throw(ArgumentError("out must be the same size as x"))             # filt.jl:531
    copyto!(out, filt(b, x))
end

# conv(u, v; algorithm) / conv!(out, u, v; algorithm)   dspbase.jl:709-792 (vectors; :direct for small / integer inputs goes to the N-d entry)
function conv(u::AbstractVector{Tu}, v::AbstractVector{Tv}; algorithm=:auto) where {Tu<:Number,Tv<:Number}
    T = promote_type(Tu, Tv)
    W = fftintype(T)
    nu, nv = length(u), length(v)
    (nu == 0 || nv == 0) && return zeros(T, max(nu + nv - 1, 0))
    alg = algorithm
    alg === :auto && (alg = T <: Union{Float32,Float64,ComplexF32,ComplexF64} ? :fast : :direct)
    alg === :fast && (alg = nu * nv < 2^16 ? :direct : :fft)
    alg === :direct && return convnd(reshape(u, :, 1), reshape(v, :, 1), :direct)[:, 1]
    big, small = nu >= nv ? (u, v) : (v, u)
    os = optimalfftfiltlength(length(small), length(big))
    alg === :fft && (alg = os < nu + nv - 1 ? :fft_overlapsave : :fft_simple)
    alg in (:fft_overlapsave, :fft_simple) ||
        throw(ArgumentError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave"))
    nfft = alg === :fft_simple ? nextfastfft(nu + nv - 1) : os
    plan = cached_ols_plan(convert(Vector{W}, small), nfft, length(big), OLS_CONV)
    download(ols_exec!(DeviceArray{W}((nu + nv - 1,)), plan, todevice(big, W), nu + nv - 1))
end

# conv(u, v; algorithm) for arrays   dspbase.jl:709-792: _conv_kern_fft! (:611-644) / _conv_td! (:646-660) on the device.
# Julia arrays are column-major, which is the layout the C ABI takes: sizes are passed as they are.
function convnd(u::AbstractArray{Tu,N}, v::AbstractArray{Tv,N}, alg::Symbol) where {Tu<:Number,Tv<:Number,N}
    T = promote_type(Tu, Tv)
    W = T <: Union{Float32,Float64,ComplexF32,ComplexF64} ? T : (T <: Complex ? ComplexF64 : Float64)
    so = size(u) .+ size(v) .- 1
    ud, vd, out = todevice(u, W), todevice(v, W), DeviceArray{W}(so)
    su, sv = collect(Int64, size(u)), collect(Int64, size(v))
    if alg === :direct
        GC.@preserve su sv check(ccall((:mdsp_convnd_direct, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                                       ud.ptr, pointer(su), vd.ptr, pointer(sv), N, mdtype(W), out.ptr, C_NULL))
    else
        GC.@preserve su sv check(ccall((:mdsp_convnd_fft, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                                       ud.ptr, pointer(su), vd.ptr, pointer(sv), N, mdtype(W), out.ptr, C_NULL))
    end
    res = download(out)
    T === W ? res : round.(T, res)                     # integer eltypes: exact in Float64 below 2^53
end
function conv(u::AbstractArray{Tu,N}, v::AbstractArray{Tv,N}; algorithm=:auto) where {Tu<:Number,Tv<:Number,N}
    T = promote_type(Tu, Tv)
    (isempty(u) || isempty(v)) && return zeros(T, max.(size(u) .+ size(v) .- 1, 0))
    alg = algorithm
    alg === :auto && (alg = T <: Union{Float32,Float64,ComplexF32,ComplexF64} ? :fast : :direct)
    alg === :fast && (alg = length(u) * length(v) < 2^16 ? :direct : :fft)
    alg in (:direct, :fft, :fft_simple, :fft_overlapsave) ||
        throw(ArgumentError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave"))
    convnd(u, v, alg === :direct ? :direct : :fft)
end
# rank promotion with trailing singleton dimensions   dspbase.jl:784-792
function conv(A::AbstractArray{<:Number,M}, B::AbstractArray{<:Number,N}; kwargs...) where {M,N}
    M < N ? conv(reshape(A, size(A)..., ntuple(_ -> 1, N - M)...), B; kwargs...) :
            conv(A, reshape(B, size(B)..., ntuple(_ -> 1, M - N)...); kwargs...)
end
# separable 2-d kernel   dspbase.jl:801-818
conv(u::AbstractVector, v::AbstractMatrix{<:Number}, A::AbstractMatrix) = size(v, 1) == 1 ? conv(reshape(u, :, 1) * v, A) :
    throw(ArgumentError("conv(u, v', A): v must be a row vector"))
function conv!(out::AbstractArray, u::AbstractArray, v::AbstractArray; algorithm=:auto)          # dspbase.jl:709-750
    res = conv(u, v; algorithm)
    size(out) == size(res) || throw(ArgumentError("output size $(size(out)) must equal input sizes minus one: $(size(res))"))   # :754
    copyto!(out, res)
end

# xcorr(u, v; padmode, scaling)   dspbase.jl:867-898
function xcorr(u::AbstractVector, v::AbstractVector=u; padmode::Symbol=:none, scaling::Symbol=:none)
    su, sv = length(u), length(v)
    scaling === :biased && su != sv && throw(DimensionMismatch("scaling only valid for vectors of same length"))
    padmode in (:none, :longest) || throw(ArgumentError("padmode keyword argument must be either :none or :longest"))
    if padmode === :longest
        n = max(su, sv)
        u, v = vcat(u, zeros(eltype(u), n - su)), vcat(v, zeros(eltype(v), n - sv))
    end
    res = conv(u, conj.(reverse(v)))
    scaling === :biased ? res ./ su : res
end

# hilbert(x)   util.jl:31-87: analytic signal of every real column
function hilbert(x::Union{AbstractVecOrMat{T},DeviceArray{T}}) where {T<:Real}
    S = fftintype(T)
    xd = todevice(x, S)
    n = size(xd, 1)
    out = DeviceArray{fftouttype(S)}(size(xd))
    n > 0 && check(ccall((:mdsp_hilbert, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                         xd.ptr, n, ncolumns(xd), n, mdtype(S), out.ptr, n, C_NULL))
    back(out, x)
end

# ---------------------------------------------------------------------------------------------- periodograms
# result types   periodograms.jl:262-330, :765-793
abstract type TFR{T} end
struct Periodogram{T,F<:AbstractVector,V<:AbstractVector{T}} <: TFR{T}
    power::V
    freq::F
end
struct Spectrogram{T,F<:AbstractVector,M<:AbstractMatrix{T}} <: TFR{T}
    power::M
    freq::F
    time::StepRangeLen{Float64,Base.TwicePrecision{Float64},Base.TwicePrecision{Float64}}
end
power(p::TFR) = p.power
freq(p::TFR) = p.freq
Base.time(p::Spectrogram) = p.time
# AbstractFFTs.rfftfreq / fftfreq (the reference's `freq` fields are AbstractFFTs.Frequencies; these are the same values)
rfftfreq(n::Integer, fs::Real=1) = (0:(n >> 1)) .* (fs / n)
fftfreq(n::Integer, fs::Real=1) = vcat(0:((n - 1) >> 1), -(n >> 1):-1) .* (fs / n)

compute_window(::Nothing, n::Int) = (nothing, Float64(n))                               # periodograms.jl:248-257
function compute_window(window::Function, n::Int)
    win = window(n)::Vector{Float64}
    (win, sum(abs2, win))
end
function compute_window(window::AbstractVector, n::Int)
    length(window) == n || throw(DimensionMismatch("length of window must match input"))
    w = convert(Vector{Float64}, window)
    (w, sum(abs2, w))
end
winptr(::Nothing) = Ptr{Cdouble}(C_NULL)
winptr(w::Vector{Float64}) = pointer(w)

# arraysplit(s, n, noverlap, nfft = n, window = nothing)   periodograms.jl:32-137: frame k = s[(k-1)(n-noverlap)+1 ...] .* window, zero
# padded to nfft.  The frames are materialised on the device by the library's framer (what the fused kernels do on the fly; bit-exact
# with the reference, tests/test_gpu_parity.py::test_frames_bit_exact) and indexed like the reference's ArraySplit.
struct ArraySplit{T} <: AbstractVector{Vector{T}}
    frames::Matrix{T}
end
Base.size(a::ArraySplit) = (size(a.frames, 2),)
Base.getindex(a::ArraySplit, k::Int) = a.frames[:, k]
function arraysplit(s::AbstractVector{T}, n::Integer, noverlap::Integer, nfft::Integer=n, window=nothing) where {T<:Number}
    (0 <= noverlap < n) || throw(DomainError((; noverlap, n), "noverlap must be between zero and n"))       # :44
    nfft >= n || throw(DomainError((; nfft, n), "nfft must be >= n"))                                         # :45
    S = fftintype(T)
    win, _ = compute_window(window, Int(n))
    sd = todevice(s, S)
    k = framecount(length(s), n, noverlap)
    out = DeviceArray{S}((Int(nfft), k))
    GC.@preserve win (k > 0 && check(ccall((:mdsp_frames, lib), Cint,
        (Ptr{Cvoid}, Int64, Cint, Int64, Int64, Int64, Ptr{Cdouble}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
        sd.ptr, length(s), mdtype(S), n, noverlap, nfft, winptr(win), 0, k, out.ptr, C_NULL)))
    ArraySplit{S}(download(out))
end

mutable struct WelchConfig                                                                        # periodograms.jl:516-587
    h::Ptr{Cvoid}
    borrowed::Bool
    nsamples::Int; noverlap::Int; onesided::Bool; nfft::Int; fs::Float64
    freq::AbstractVector; window; r::Float64; intype::DataType; nout::Int
end
function WelchConfig(nsamples, ::Type{T}; n::Int=nsamples >> 3, noverlap::Int=n >> 1, onesided::Bool=T <: Real,
                     nfft::Int=nextfastfft(n), fs::Real=1, window=nothing, engine=ENGINE_AUTO, cached::Bool=false) where {T}
    onesided && T <: Complex && throw(ArgumentError("cannot compute one-sided FFT of a complex signal"))
    nfft >= n || throw(DomainError((; nfft, n), "nfft must be >= n"))
    win, norm2 = compute_window(window, n)
    r = fs * norm2
    S = fftintype(T)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    if cached   # the function-style welch_pgram(s, n, noverlap): plan from the library's LRU, borrowed
        GC.@preserve win check(ccall((:mdsp_welch_plan_cached, lib), Cint,
            (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Cdouble}, Cdouble, Cint, Cint, Cint, Ptr{Cvoid}),
            p, n, noverlap, nfft, winptr(win), r, onesided, mdtype(S), engine, C_NULL))
    else
        GC.@preserve win check(ccall((:mdsp_welch_plan_create, lib), Cint,
            (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Cdouble}, Cdouble, Cint, Cint, Cint),
            p, n, noverlap, nfft, winptr(win), r, onesided, mdtype(S), engine))
    end
    no, eng = Ref{Int64}(0), Ref{Cint}(0)
    check(ccall((:mdsp_welch_plan_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Cint}), p[], no, eng))
    cfg = WelchConfig(p[], cached, n, noverlap, onesided, nfft, fs, onesided ? rfftfreq(nfft, fs) : fftfreq(nfft, fs), win, r, S, Int(no[]))
    cached || finalizer(x -> ccall((:mdsp_welch_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.h), cfg)
    cfg
end
WelchConfig(data::AbstractArray; kw...) = WelchConfig(size(data, 1), eltype(data); kw...)

# welch_pgram(s, config) -> Periodogram   periodograms.jl:702-705, :746-759.  Columns of a matrix are channels (`power` is then a matrix).
function welch_power(s::DeviceArray{T}, config::WelchConfig) where {T}
    len = size(s, 1); nch = ncolumns(s)
    out = DeviceArray{fftabs2type(config.intype)}(ndims(s) == 1 ? (config.nout,) : (config.nout, nch))
    check(ccall((:mdsp_welch_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                config.h, s.ptr, len, nch, len, out.ptr, config.nout, C_NULL))
    out
end
# Host Arrays: mdsp_welch_exec_host (time chunks of whole frames through the pipeline; Float64 sums accumulate in the plan)
function welch_power(s::Array{T}, config::WelchConfig; pinned::Bool=false) where {T}
    len = size(s, 1); nch = ncolumns(s)
    out = Array{fftabs2type(T)}(undef, ndims(s) == 1 ? (config.nout,) : (config.nout, nch))
    GC.@preserve s out check(ccall((:mdsp_welch_exec_host, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Cint),
                                   config.h, pointer(s), len, nch, len, pointer(out), config.nout, pinned ? HOST_PINNED : Cint(0)))
    out
end
function welch_pgram(s::Union{AbstractVecOrMat{T},DeviceArray{T}}, config::WelchConfig; kw...) where {T<:Number}
    fftintype(T) == config.intype ||
        throw(ArgumentError("float(eltype(s)) = $T doesn't match the eltype of the input buffer: $(config.intype)."))
    p = s isa DeviceArray ? welch_power(s, config) : welch_power(convert(Array{config.intype}, s), config; kw...)
    ndims(p) == 1 ? Periodogram(p isa DeviceArray ? download(p) : p, config.freq) : (power = p, freq = config.freq)
end
welch_pgram(s::AbstractVector, n::Int=length(s) >> 3, noverlap::Int=n >> 1; kw...) =
    welch_pgram(s, WelchConfig(s; n, noverlap, cached=true, kw...))
function welch_pgram!(out::AbstractVector, s::AbstractVector{T}, config::WelchConfig) where {T<:Number}        # periodograms.jl:734-744
    length(out) == length(config.freq) || throw(DimensionMismatch("Expected `output` to be of length `length(config.freq)`; got `length(output)` = $(length(out)) and `length(config.freq)` = $(length(config.freq))"))
    eltype(out) == fftabs2type(T) || throw(ArgumentError("Eltype of output ($(eltype(out))) doesn't match the expected type: $(fftabs2type(T))."))
    float(T) == config.intype || throw(ArgumentError("float(eltype(s)) = $T doesn't match the eltype of the input buffer: $(config.intype)."))
    copyto!(out, power(welch_pgram(s, config)))
    Periodogram(out, config.freq)
end
welch_pgram!(out::AbstractVector, s::AbstractVector, n::Int=length(s) >> 3, noverlap::Int=n >> 1; kw...) =
    welch_pgram!(out, s, WelchConfig(s; n, noverlap, cached=true, kw...))

# Streaming form (mdsp_welch_exec IS reset + accumulate + finalize): a stream handed over slice by slice -- each slice whole frames,
# consecutive slices overlapping by n - hop samples -- or by several ranks (welch_allreduce!).   periodograms.jl:746-759
welch_reset!(c::WelchConfig) = (check(ccall((:mdsp_welch_reset, lib), Cint, (Ptr{Cvoid},), c.h)); c)
function welch_accumulate!(c::WelchConfig, s::DeviceArray)
    len = size(s, 1)
    check(ccall((:mdsp_welch_accumulate, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}), c.h, s.ptr, len, ncolumns(s), len, C_NULL))
    c
end
function welch_frames_accumulated(c::WelchConfig)
    k = Ref{Int64}(0)
    check(ccall((:mdsp_welch_frames_accumulated, lib), Cint, (Ptr{Cvoid}, Ref{Int64}), c.h, k)); Int(k[])
end
# the Float64 accumulator itself (device pointer, element count): what a caller-side collective (MPI.jl ...) would sum over ranks
function welch_accumulator(c::WelchConfig)
    p, n = Ref{Ptr{Cvoid}}(C_NULL), Ref{Int64}(0)
    check(ccall((:mdsp_welch_accumulator, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), c.h, p, n))
    (ptr = p[], count = Int(n[]))
end
function welch_finalize(c::WelchConfig, nch::Integer=1; frames_total::Integer=0)
    out = DeviceArray{fftabs2type(c.intype)}((c.nout, Int(nch)))
    check(ccall((:mdsp_welch_finalize, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}), c.h, frames_total, out.ptr, c.nout, C_NULL))
    out
end

# ---------------------------------------------------------------------------------------------- multi-GPU (one process per GPU)
# The reference loops over columns / channels serially (Filters/filt.jl:504; `mapslices` stream_filt.jl:768).  Here rank r owns a
# contiguous block of channels and runs the single-GPU methods on them; the ONLY collective is the all-reduce of an nout-value PSD for
# the cross-channel Welch mean -- RCCL over xGMI, behind the C ABI.  Bootstrap: rank 0 makes the 128-byte id, the host ships it
# (e.g. `MPI.Bcast!(id, 0, comm)` or a shared file), every rank calls Comm(id, rank, nranks) after init(device).
mutable struct Comm
    h::Ptr{Cvoid}
    rank::Int
    nranks::Int
end
function unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id check(ccall((:mdsp_comm_unique_id, lib), Cint, (Ptr{Cvoid},), pointer(id)))
    id
end
function Comm(id::Vector{UInt8}, rank::Integer, nranks::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve id check(ccall((:mdsp_comm_init_rank, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Cint, Cint), p, pointer(id), rank, nranks))
    r, n = Ref{Cint}(-1), Ref{Cint}(-1)
    check(ccall((:mdsp_comm_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), p[], r, n))
    c = Comm(p[], Int(r[]), Int(n[]))
    finalizer(x -> ccall((:mdsp_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
    c
end
channel_shard(nch::Integer, rank::Integer, nranks::Integer) = (per = cld(nch, nranks); (min(rank * per, nch) + 1):min((rank + 1) * per, nch))
# in-place sum over ranks of a device array (the bare transport)
function allreduce_sum!(x::DeviceArray{T}, comm::Comm) where {T<:Union{Float32,Float64}}
    check(ccall((:mdsp_allreduce_sum, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}), comm.h, x.ptr, length(x), mdtype(T), C_NULL)); x
end
# sum over this rank's channels (columns) of a (len, nch) device matrix
function channel_sum(x::DeviceArray{T}) where {T<:Union{Float32,Float64}}
    len = size(x, 1)
    out = DeviceArray{T}((len,))
    check(ccall((:mdsp_channel_sum, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}), x.ptr, len, ncolumns(x), len, mdtype(T), out.ptr, C_NULL))
    out
end

# mean over ALL channels (across ranks) of the per-channel Welch PSDs: local sum on the device -> ncclAllReduce(sum) of nout values ->
# 1/nch_total, one C-ABI call.  `s`: this rank's channels as columns.
function welch_channel_mean(s::DeviceArray{T}, config::WelchConfig, nch_total::Integer, comm::Union{Comm,Nothing}=nothing) where {T}
    len = size(s, 1); nloc = ncolumns(s)
    R = fftabs2type(config.intype)
    psd = DeviceArray{R}((config.nout, max(nloc, 1)))
    nloc > 0 && check(ccall((:mdsp_welch_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                            config.h, s.ptr, len, nloc, len, psd.ptr, config.nout, C_NULL))
    mean = DeviceArray{R}((config.nout,))
    check(ccall((:mdsp_welch_mean_allreduce, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                config.h, psd.ptr, nloc, config.nout, nch_total, mean.ptr, comm === nothing ? C_NULL : comm.h, C_NULL))
    mean
end
# one stream split along time over ranks: after welch_accumulate! on every rank's slice, sum the Float64 accumulators and frame counts
welch_allreduce!(c::WelchConfig, comm::Comm) =
    (check(ccall((:mdsp_welch_allreduce, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), c.h, comm.h, C_NULL)); c)

# ---------------------------------------------------------------------------------------------- stft / spectrogram / periodogram
# periodograms.jl:872-897, :828-837, :393-417.  The plan (window upload, tables, transforms) comes from the library's LRU: the reference
# builds its FFTW plan on every call, here a call costs ~90 us instead of ~370.
function stft_plan(n, noverlap, nfft, win, r, onesided::Bool, psdonly::Bool, ::Type{S}, engine) where {S}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve win check(ccall((:mdsp_stft_plan_cached, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Cdouble}, Cdouble, Cint, Cint, Cint, Cint, Ptr{Cvoid}),
        p, n, noverlap, nfft, winptr(win), r, onesided, psdonly, mdtype(S), engine, C_NULL))
    no, eng = Ref{Int64}(0), Ref{Cint}(0)
    check(ccall((:mdsp_stft_plan_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Cint}), p[], no, eng))
    (p[], Int(no[]))
end
# an owned plan (mdsp_stft_plan_create / _destroy) for callers that keep one across many calls
mutable struct StftPlan
    h::Ptr{Cvoid}
    function StftPlan(n, noverlap, nfft, win, r, onesided::Bool, psdonly::Bool, ::Type{S}, engine=ENGINE_AUTO) where {S}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve win check(ccall((:mdsp_stft_plan_create, lib), Cint,
            (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Cdouble}, Cdouble, Cint, Cint, Cint, Cint),
            p, n, noverlap, nfft, winptr(win), r, onesided, psdonly, mdtype(S), engine))
        o = new(p[])
        finalizer(x -> ccall((:mdsp_stft_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.h), o)
        o
    end
end
function stft(s::Union{AbstractVecOrMat{T},DeviceArray{T}}, n::Int=size(s, 1) >> 3, noverlap::Int=n >> 1, psdonly::Bool=false;
              onesided::Bool=T <: Real, nfft::Int=nextfastfft(n), fs::Real=1, window=nothing, engine=ENGINE_AUTO, pinned::Bool=false) where {T}
    onesided && T <: Complex && throw(ArgumentError("cannot compute one-sided FFT of a complex signal"))
    win, norm2 = compute_window(window, n)
    (0 ≤ noverlap < n) || throw(DomainError((; noverlap, n), "noverlap must be between zero and n"))
    nfft >= n || throw(DomainError((; nfft, n), "nfft must be >= n"))
    S = fftintype(T)
    plan, nout = stft_plan(n, noverlap, nfft, win, fs * norm2, onesided, psdonly, S, engine)
    len = size(s, 1); nch = ncolumns(s)
    k = framecount(len, n, noverlap)
    O = psdonly ? fftabs2type(S) : fftouttype(S)
    dims = ndims(s) == 1 ? (nout, k) : (nout, k, nch)
    if s isa DeviceArray
        sd = todevice(s, S)
        out = DeviceArray{O}(dims)
        k > 0 && check(ccall((:mdsp_stft_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                             plan, sd.ptr, len, nch, len, out.ptr, nout, nout * k, C_NULL))
        return out
    end
    # host Arrays: mdsp_stft_exec_host -- the output (2-8x the input) never lives on the device whole
    sh = convert(Array{S}, s)
    out = zeros(O, dims)
    GC.@preserve sh out (k > 0 && check(ccall((:mdsp_stft_exec_host, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Cint),
                                              plan, pointer(sh), len, nch, len, pointer(out), nout, nout * k, pinned ? HOST_PINNED : Cint(0))))
    out
end
function spectrogram(s::AbstractVector{T}, n::Int=length(s) >> 3, noverlap::Int=n >> 1; onesided::Bool=T <: Real,
                     nfft::Int=nextfastfft(n), fs::Real=1, window=nothing) where {T}
    out = stft(s, n, noverlap, true; onesided, nfft, fs, window)
    Spectrogram(out, onesided ? rfftfreq(nfft, fs) : fftfreq(nfft, fs),
                (n / 2 : n - noverlap : (size(out, 2) - 1) * (n - noverlap) + n / 2) / fs)                    # :835
end
function periodogram(s::AbstractVector{T}; onesided::Bool=T <: Real, nfft::Int=nextfastfft(length(s)), fs::Real=1, window=nothing) where {T}
    onesided && T <: Complex && throw(ArgumentError("cannot compute one-sided FFT of a complex signal"))     # :396
    nfft >= length(s) || throw(DomainError((; nfft, n=length(s)), "nfft must be >= n = length(s)"))        # :397
    p = stft(s, length(s), 0, true; onesided, nfft, fs, window)
    Periodogram(vec(p), onesided ? rfftfreq(nfft, fs) : fftfreq(nfft, fs))
end

# ---------------------------------------------------------------------------------------------- filter design needed as INPUTS of the path
# kaiserord / Kaiser window / windowed-sinc low-pass / resample_filter   Filters/design.jl:547-559, :235-240, :598-602, :683-720; windows.jl
function kaiserord(transitionwidth::Real, attenuation::Real=60)
    n = ceil(Int, (attenuation - 7.95) / (π * 2.285 * transitionwidth)) + 1
    β = attenuation > 50 ? 0.1102 * (attenuation - 8.7) :
        attenuation >= 21 ? 0.5842 * (attenuation - 21)^0.4 + 0.07886 * (attenuation - 21) : 0.0
    (n, β / π)
end
function besseli0(x::Float64)          # power series of the modified Bessel function I0 (enough for Kaiser windows: |x| < 50)
    s = t = 1.0
    for k in 1:200
        t *= (x / (2k))^2
        s += t
        t < eps(s) && break
    end
    s
end
kaiser(n::Integer, α::Real) = n == 1 ? [1.0] : [besseli0(π * α * sqrt(1 - (2 * (k - 1) / (n - 1) - 1)^2)) / besseli0(π * α) for k in 1:n]
function lowpass_firwindow(w::Real, window::Vector{Float64}; fs::Real=2, scale::Bool=true)
    w > 0 || throw(DomainError(w, "frequencies must be positive"))
    f = 2 * w / fs
    f < 1 || throw(DomainError(w, "frequencies must be less than the Nyquist frequency"))
    n = length(window)
    sincn(x) = x == 0 ? 1.0 : sin(π * x) / (π * x)
    taps = [f * sincn(f * (k - (n + 1) / 2)) * window[k] for k in 1:n]
    scale ? taps ./ sum(taps) : taps
end
function resample_filter(rate::AbstractFloat, Nϕ::Integer=32, rel_bw::Real=1.0, attenuation::Real=60)        # design.jl:683-702
    f_nyq = rate >= 1.0 ? 1.0 / Nϕ : rate / Nϕ
    cutoff = f_nyq * rel_bw
    hLen, α = kaiserord(cutoff * 0.2, attenuation)
    hLen = Nϕ * ceil(Int, hLen / Nϕ)
    iseven(hLen) && (hLen += 1)
    lowpass_firwindow(cutoff, kaiser(hLen, α)) .* Nϕ
end
function resample_filter(rate::Union{Integer,Rational}, rel_bw::Real=1.0, attenuation::Real=60)               # design.jl:704-720
    Nϕ = numerator(rate)
    f_nyq = min(1 / Nϕ, 1 / denominator(rate))
    cutoff = f_nyq * rel_bw
    hLen, α = kaiserord(cutoff * 0.2, attenuation)
    hLen = Nϕ * ceil(Int, hLen / Nϕ)
    iseven(hLen) && (hLen += 1)
    lowpass_firwindow(cutoff, kaiser(hLen, α)) .* Nϕ
end

# ---------------------------------------------------------------------------------------------- FIRFilter / resample
mutable struct FIRFilter                                                                 # stream_filt.jl:137-178
    h::Ptr{Cvoid}
    taps::Vector
    ratio::Rational{Int}
    xtype::DataType
    nch::Int
end
# exact=true: the generic kernel only -- every output reads exactly its own tapsPerϕ-sample window, so NaN / Inf samples leave exactly DSP.jl's hole
function FIRFilter(taps::Vector{Th}, ratio::Union{Integer,Rational}=1; xtype::DataType=Th, nch::Integer=1, exact::Bool=false) where {Th<:Union{Float32,Float64}}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    r = convert(Rational{Int}, ratio)
    GC.@preserve taps check(ccall((:mdsp_fir_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Int64),
                                  p, pointer(taps), length(taps), numerator(r), denominator(r), mdtype(Th), mdtype(xtype), nch))
    exact && check(ccall((:mdsp_fir_set_exact, lib), Cint, (Ptr{Cvoid}, Cint), p[], 1))
    f = FIRFilter(p[], taps, r, xtype, nch)
    finalizer(x -> ccall((:mdsp_fir_destroy, lib), Cint, (Ptr{Cvoid},), x.h), f)
    f
end
reset!(f::FIRFilter) = (check(ccall((:mdsp_fir_reset, lib), Cint, (Ptr{Cvoid},), f.h)); f)               # :247-276
setphase!(f::FIRFilter, ϕ::Real) = check(ccall((:mdsp_fir_setphase, lib), Cint, (Ptr{Cvoid}, Cdouble), f.h, ϕ))  # :216-229
function timedelay(f::FIRFilter)                                                                           # :400-403
    τ = Ref{Cdouble}(0)
    check(ccall((:mdsp_fir_timedelay, lib), Cint, (Ptr{Cvoid}, Ref{Cdouble}), f.h, τ)); τ[]
end
function firinfo(f::FIRFilter)
    kind, od = Ref{Cint}(0), Ref{Cint}(0)
    L, M, tp, hl = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:mdsp_fir_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Cint}), f.h, kind, L, M, tp, hl, od))
    (kind = Int(kind[]), L = Int(L[]), M = Int(M[]), tapsPerϕ = Int(tp[]), historyLen = Int(hl[]), outtype = JLTYPE[od[] + 1])
end
# the reference's state: (ϕIdx, inputDeficit, history) with 1-based ϕIdx   stream_filt.jl:141, :59-79
function getstate(f::FIRFilter)
    hl = firinfo(f).historyLen
    hist = zeros(f.xtype, max(hl, 1), f.nch)
    ϕ, d = Ref{Int64}(0), Ref{Int64}(0)
    GC.@preserve hist check(ccall((:mdsp_fir_get_state, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ptr{Cvoid}), f.h, ϕ, d, pointer(hist)))
    (ϕIdx = Int(ϕ[]), inputDeficit = Int(d[]), history = hist[1:hl, :])
end
function setstate!(f::FIRFilter, ϕIdx::Integer, inputDeficit::Integer, history::Union{Nothing,AbstractMatrix}=nothing)
    hist = history === nothing ? nothing : convert(Matrix{f.xtype}, history)
    GC.@preserve hist check(ccall((:mdsp_fir_set_state, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}), f.h, ϕIdx, inputDeficit,
                                  hist === nothing ? C_NULL : pointer(hist)))
    f
end
# which device kernel a chunk of `n` samples would take: 0 generic, 1 register-tap, 2 matrix-core (diagnostics; results do not depend on it)
function kernel_path(f::FIRFilter, n::Integer)
    p = Ref{Cint}(-1)
    check(ccall((:mdsp_fir_kernel_path, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Cint}), f.h, n, p)); Int(p[])
end
# geometry of the matrix-core kernel for a filter of hlen taps at L // M (host arithmetic only; tests place their seam windows with it)
function fir_mm_geometry(L::Integer, M::Integer, hlen::Integer, ::Type{Th}, ::Type{Tx}) where {Th,Tx}
    o = zeros(Int64, 12)
    GC.@preserve o check(ccall((:mdsp_fir_mm_geometry, lib), Cint, (Int64, Int64, Int64, Cint, Cint, Ptr{Int64}), L, M, hlen, mdtype(Th), mdtype(Tx), pointer(o)))
    o
end
function outputlength(f::FIRFilter, inlen::Integer)                                                        # :324-338
    o = Ref{Int64}(0)
    check(ccall((:mdsp_fir_outputlength, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Int64}), f.h, inlen, o)); Int(o[])
end
function inputlength(f::FIRFilter, outlen::Integer, r::RoundingMode=RoundDown)                             # :366-383
    o = Ref{Int64}(0)
    check(ccall((:mdsp_fir_inputlength, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ref{Int64}), f.h, outlen,
                (r == RoundUp || r == RoundFromZero) ? 1 : 0, o)); Int(o[])
end

# filt(f, x): next chunk, state carried on the device   stream_filt.jl:627-637
function filt(f::FIRFilter, x::DeviceArray)
    xd = todevice(x, f.xtype)
    xlen = size(xd, 1)
    ycap = max(outputlength(f, xlen), 0)
    y = DeviceArray{firinfo(f).outtype}(f.nch == 1 ? (ycap,) : (ycap, f.nch))
    nw = Ref{Int64}(0)
    check(ccall((:mdsp_fir_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ref{Int64}, Ptr{Cvoid}),
                f.h, xd.ptr, xlen, xlen, y.ptr, ycap, max(ycap, 1), nw, C_NULL))
    nw[] == ycap || throw(AssertionError("Length of resampled output different from expectation."))
    y
end
# host Arrays: mdsp_fir_exec_host -- the stream passes through the stateful filter in time chunks (H2D || kernel || D2H)
function filt(f::FIRFilter, x::AbstractVecOrMat; pinned::Bool=false)
    xh = convert(Array{f.xtype}, x)
    xlen = size(xh, 1)
    ycap = max(outputlength(f, xlen), 0)
    y = Array{firinfo(f).outtype}(undef, f.nch == 1 ? (ycap,) : (ycap, f.nch))
    nw = Ref{Int64}(0)
    GC.@preserve xh y check(ccall((:mdsp_fir_exec_host, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ref{Int64}, Cint),
                                  f.h, pointer(xh), xlen, xlen, pointer(y), ycap, max(ycap, 1), nw, pinned ? HOST_PINNED : Cint(0)))
    nw[] == ycap || throw(AssertionError("Length of resampled output different from expectation."))
    y
end
# stateless filt(h, x, ratio)   stream_filt.jl:663-666
filt(h::Vector, x::AbstractVector, ratio::Union{Integer,Rational}) = filt(FIRFilter(h, ratio; xtype=fftintype(eltype(x))), x)

# resample(x, rate[, h]; dims)   stream_filt.jl:688-725 (vector) / :747-775 (array: every slice along `dims` is a channel)
function resample(x::AbstractVector{T}, rate::Union{Integer,Rational}, h::Vector=resample_filter(rate)) where {T}
    S = fftintype(T)
    f = FIRFilter(convert(Vector{real(promote_type(eltype(h), S)) == Float32 ? Float32 : Float64}, h), rate; xtype=S)
    rate == 1 || setphase!(f, timedelay(f))                              # undelay!, :706-714
    outLen = ceil(Int, length(x) * rate)
    xp = zeros(S, inputlength(f, outLen, RoundUp)); xp[1:length(x)] .= x # _zeropad, :699
    y = filt(f, xp)
    length(y) >= outLen || throw(AssertionError("Resample output shorter than expected."))                  # :722
    y[1:outLen]
end
function resample(x::AbstractArray{T}, rate::Union{Integer,Rational}, h::Vector=resample_filter(rate); dims::Integer) where {T}
    S = fftintype(T)
    xm = reshape(permutedims(x, (dims, setdiff(1:ndims(x), dims)...)), size(x, dims), :)      # slices along `dims` as columns
    nch = size(xm, 2)
    f = FIRFilter(convert(Vector{real(promote_type(eltype(h), S)) == Float32 ? Float32 : Float64}, h), rate; xtype=S, nch)
    rate == 1 || setphase!(f, timedelay(f))
    outLen = ceil(Int, size(xm, 1) * rate)
    xp = zeros(S, inputlength(f, outLen, RoundUp), nch); xp[1:size(xm, 1), :] .= xm
    y = filt(f, xp)
    size(y, 1) >= outLen || throw(AssertionError("Resample output shorter than expected."))
    osz = (outLen, (size(x, d) for d in setdiff(1:ndims(x), dims))...)
    permutedims(reshape(y[1:outLen, :], osz), invperm((dims, setdiff(1:ndims(x), dims)...)))
end

# ------------------------------------------------------------------------------ FIRArbitrary (floating-point rate)
mutable struct FIRArbitraryFilter                                                        # stream_filt.jl:92-156
    h::Ptr{Cvoid}
    taps::Vector
    rate::Float64
    Nϕ::Int
    xtype::DataType
    nch::Int
end
function FIRFilter(taps::Vector{Th}, rate::AbstractFloat, Nϕ::Integer=32; xtype::DataType=Th, nch::Integer=1) where {Th<:Union{Float32,Float64}}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve taps check(ccall((:mdsp_firarb_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Cdouble, Int64, Cint, Cint, Int64),
                                  p, pointer(taps), length(taps), Float64(rate), Nϕ, mdtype(Th), mdtype(xtype), nch))
    f = FIRArbitraryFilter(p[], taps, Float64(rate), Nϕ, xtype, nch)
    finalizer(x -> ccall((:mdsp_firarb_destroy, lib), Cint, (Ptr{Cvoid},), x.h), f)
    f
end
reset!(f::FIRArbitraryFilter) = (check(ccall((:mdsp_firarb_reset, lib), Cint, (Ptr{Cvoid},), f.h)); f)            # :260-276
setphase!(f::FIRArbitraryFilter, ϕ::Real) = check(ccall((:mdsp_firarb_setphase, lib), Cint, (Ptr{Cvoid}, Cdouble), f.h, ϕ))  # :231-239
function timedelay(f::FIRArbitraryFilter)                                                                          # :400-401
    τ = Ref{Cdouble}(0)
    check(ccall((:mdsp_firarb_timedelay, lib), Cint, (Ptr{Cvoid}, Ref{Cdouble}), f.h, τ)); τ[]
end
function firinfo(f::FIRArbitraryFilter)
    nϕ, tp, hl, od, Δ = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0), Ref{Cdouble}(0)
    check(ccall((:mdsp_firarb_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Cint}, Ref{Cdouble}), f.h, nϕ, tp, hl, od, Δ))
    (Nϕ = Int(nϕ[]), tapsPerϕ = Int(tp[]), historyLen = Int(hl[]), outtype = JLTYPE[od[] + 1], Δ = Δ[])
end
# the reference's state (stream_filt.jl:96-104): ϕAccumulator, α, ϕIdx, inputDeficit, xIdx, history
function getstate(f::FIRArbitraryFilter)
    hl = firinfo(f).historyLen
    hist = zeros(f.xtype, max(hl, 1), f.nch)
    acc, α = Ref{Cdouble}(0), Ref{Cdouble}(0)
    ϕ, d, xi = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
    GC.@preserve hist check(ccall((:mdsp_firarb_get_state, lib), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ptr{Cvoid}),
                                  f.h, acc, α, ϕ, d, xi, pointer(hist)))
    (ϕAccumulator = acc[], α = α[], ϕIdx = Int(ϕ[]), inputDeficit = Int(d[]), xIdx = Int(xi[]), history = hist[1:hl, :])
end
function setstate!(f::FIRArbitraryFilter, ϕAccumulator::Real, inputDeficit::Integer, history::Union{Nothing,AbstractMatrix}=nothing)
    hist = history === nothing ? nothing : convert(Matrix{f.xtype}, history)
    GC.@preserve hist check(ccall((:mdsp_firarb_set_state, lib), Cint, (Ptr{Cvoid}, Cdouble, Int64, Ptr{Cvoid}), f.h, ϕAccumulator, inputDeficit,
                                  hist === nothing ? C_NULL : pointer(hist)))
    f
end
# trajectories evaluated by the device scan / by the serial host loop so far (diagnostics)
function scan_stats(f::FIRArbitraryFilter)
    a, b = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:mdsp_firarb_scan_stats, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), f.h, a, b))
    (scanned = Int(a[]), serial = Int(b[]))
end
function outputlength(f::FIRArbitraryFilter, inlen::Integer)                                                       # :340-342
    o = Ref{Int64}(0)
    check(ccall((:mdsp_firarb_outputlength, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Int64}), f.h, inlen, o)); Int(o[])
end
function inputlength(f::FIRArbitraryFilter, outlen::Integer, r::RoundingMode=RoundDown)                            # :385-389
    o = Ref{Int64}(0)
    check(ccall((:mdsp_firarb_inputlength, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ref{Int64}), f.h, outlen,
                (r == RoundUp || r == RoundFromZero) ? 1 : 0, o)); Int(o[])
end
# filt(f, x): buffer of outputlength + 1 samples, resized to samplesWritten (allocate_output :639-655, filt :627-637)
function filt(f::FIRArbitraryFilter, x::Union{AbstractVecOrMat,DeviceArray})
    xd = todevice(x, f.xtype)
    xlen = size(xd, 1)
    ycap = max(outputlength(f, xlen), 0) + 1
    y = DeviceArray{firinfo(f).outtype}(f.nch == 1 ? (ycap,) : (ycap, f.nch))
    nw = Ref{Int64}(0)
    check(ccall((:mdsp_firarb_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ref{Int64}, Ptr{Cvoid}),
                f.h, xd.ptr, xlen, xlen, y.ptr, ycap, ycap, nw, C_NULL))
    yh = back(y, x)
    yh isa DeviceArray ? yh : (ndims(yh) == 1 ? yh[1:nw[]] : yh[1:nw[], :])
end
filt(h::Vector, x::AbstractVector, rate::AbstractFloat, Nϕ::Integer=32) = filt(FIRFilter(h, rate, Nϕ; xtype=fftintype(eltype(x))), x)   # :669-672
# resample(x, rate::AbstractFloat[, h, Nϕ = 32])   stream_filt.jl:692-694, :752-755
function resample(x::AbstractVecOrMat{T}, rate::AbstractFloat, h::Vector=resample_filter(rate), Nϕ::Integer=32) where {T}
    S = fftintype(T)
    nch = size(x, 2)
    f = FIRFilter(convert(Vector{real(promote_type(eltype(h), S)) == Float32 ? Float32 : Float64}, h), rate, Nϕ; xtype=S, nch)
    setphase!(f, timedelay(f))                                          # undelay!
    outLen = ceil(Int, size(x, 1) * rate)
    npad = inputlength(f, outLen, RoundUp)
    xp = zeros(S, npad, nch); xp[1:size(x, 1), :] .= x
    y = filt(f, ndims(x) == 1 ? vec(xp) : xp)
    size(y, 1) >= outLen || throw(AssertionError("Resample output shorter than expected."))   # :722
    ndims(x) == 1 ? y[1:outLen] : y[1:outLen, :]
end
# the serial phase-accumulator recurrence on the host and its parallel evaluation (tests compare the two bit for bit)
function arb_trajectory(ϕAcc::Real, inputDeficit::Integer, rate::Real, Nϕ::Integer, xlen::Integer, block::Integer, cap::Integer)
    ax, aa = zeros(Int64, cap), zeros(Float64, cap)
    nout, dend = Ref{Int64}(0), Ref{Int64}(0)
    aend = Ref{Cdouble}(0)
    GC.@preserve ax aa check(ccall((:mdsp_arb_trajectory, lib), Cint,
        (Cdouble, Int64, Cdouble, Int64, Int64, Int64, Ptr{Int64}, Ptr{Cdouble}, Int64, Ref{Int64}, Ref{Cdouble}, Ref{Int64}),
        ϕAcc, inputDeficit, rate, Nϕ, xlen, block, pointer(ax), pointer(aa), cap, nout, aend, dend))
    (anchors_x = ax, anchors_acc = aa, nout = Int(nout[]), ϕAcc_end = aend[], deficit_end = Int(dend[]))
end

# the host emulation of the device's parallel evaluation of the same recurrence (used == false: the scan does not apply to these arguments)
function arb_trajectory_scan(ϕAcc::Real, inputDeficit::Integer, rate::Real, Nϕ::Integer, xlen::Integer, pilot::Integer, cap::Integer)
    ax, aa = zeros(Int64, cap), zeros(Float64, cap)
    nout, dend = Ref{Int64}(0), Ref{Int64}(0)
    aend = Ref{Cdouble}(0)
    used, passes = Ref{Cint}(0), Ref{Cint}(0)
    GC.@preserve ax aa check(ccall((:mdsp_arb_trajectory_scan, lib), Cint,
        (Cdouble, Int64, Cdouble, Int64, Int64, Int64, Ptr{Int64}, Ptr{Cdouble}, Int64, Ref{Int64}, Ref{Cdouble}, Ref{Int64}, Ref{Cint}, Ref{Cint}),
        ϕAcc, inputDeficit, rate, Nϕ, xlen, pilot, pointer(ax), pointer(aa), cap, nout, aend, dend, used, passes))
    (used = used[] != 0, passes = Int(passes[]), anchors_x = ax, anchors_acc = aa, nout = Int(nout[]), ϕAcc_end = aend[], deficit_end = Int(dend[]))
end
# updates at which the device replay's branch-free form of update! (stream_filt.jl:567-577) would differ from the reference form: always 0
function arb_replay_check(ϕAcc::Real, rate::Real, Nϕ::Integer, nsteps::Integer)
    m = Ref{Int64}(0)
    check(ccall((:mdsp_arb_replay_check, lib), Cint, (Cdouble, Cdouble, Int64, Int64, Ref{Int64}), ϕAcc, rate, Nϕ, nsteps, m)); Int(m[])
end

# ---------------------------------------------------------------------------------------------- DF2TFilter (FIR) and filtfilt
# DF2TFilter(PolynomialRatio(b, [1]))   Filters/filt.jl:122-181: the TDF-II register file `state` (length(b) - 1, columns) is carried between calls
mutable struct DF2TFilter{T}
    b::Vector{T}
    state::Matrix{T}
end
function DF2TFilter(b::AbstractVector{<:Real}, a::Union{Real,AbstractVector{<:Real}}=1.0; coldims::Integer=1)
    a1 = a isa Real ? a : (length(a) == 1 ? a[1] : throw(UnsupportedError("IIR DF2TFilter is a serial recursion; only FIR coefficients run on the device")))
    isempty(b) && throw(ArgumentError("filter coefficients must be non-empty"))
    a1 == 0 && throw(ArgumentError("filter vector a[1] must be nonzero"))
    T = eltype(b) == Float32 ? Float32 : Float64
    taps = convert(Vector{T}, a1 == 1 ? b : b ./ a1)                                    # PolynomialRatio normalises by a[1]
    DF2TFilter{T}(taps, zeros(T, length(taps) - 1, coldims))
end
function filt(f::DF2TFilter{T}, x::AbstractVecOrMat{Tx}) where {T,Tx<:Real}
    size(x, 2) == size(f.state, 2) || throw(ArgumentError("state size must match x"))   # :158
    W = promote_type(T, fftintype(Tx))
    nb = length(f.b)
    nb == 1 && return convert(Array{W}, x) .* f.b[1]                                    # mul!(out, x, b[1]), :163
    xd = todevice(x, W)
    nx = size(xd, 1)
    y = DeviceArray{W}(size(xd))
    si = upload(convert(Matrix{W}, f.state))
    taps = convert(Vector{W}, f.b)
    GC.@preserve taps (nx > 0 && check(ccall((:mdsp_tdfir_state_exec, lib), Cint,
        (Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
        pointer(taps), nb, mdtype(W), xd.ptr, nx, ncolumns(xd), nx, y.ptr, nx, si.ptr, C_NULL)))
    f.state = convert(Matrix{T}, download(si))
    download(y)
end
# filtfilt(b, x) / filtfilt(b, a, x) with scalar a   filt.jl:301-338: odd-symmetric extension, one pass with conv(b, reverse(b)), trim
function filtfilt(b::AbstractVector{<:Real}, x::AbstractVecOrMat{Tx}) where {Tx<:Real}
    nb, n = length(b), size(x, 1)
    nb - 1 <= n - 1 || throw(ArgumentError("the signal must be longer than the filter order"))
    bw = convert(Vector{eltype(b) == Float32 ? Float32 : Float64}, b)
    newb = conv(bw, reverse(bw); algorithm=:direct)                                     # :309-314
    W = promote_type(eltype(bw), fftintype(Tx))
    xd = todevice(x, W)
    ext = DeviceArray{W}(ndims(x) == 1 ? (n + 2 * (nb - 1),) : (n + 2 * (nb - 1), size(x, 2)))
    n > 0 && check(ccall((:mdsp_extrapolate, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                         xd.ptr, n, ncolumns(xd), n, mdtype(W), nb - 1, ext.ptr, n + 2 * (nb - 1), C_NULL))   # extrapolate_signal!, :243-257
    y = download(filt(newb, ext))                                                       # filt!(extrapolated, newb, extrapolated), :322
    ndims(x) == 1 ? y[2nb-1:end] : y[2nb-1:end, :]                                      # drop garbage at start, :325
end
filtfilt(b::AbstractVector{<:Real}, a::Union{Real,AbstractVector{<:Real}}, x) =
    (a isa Real || length(a) == 1) ? filtfilt(b ./ (a isa Real ? a : a[1]), x) :
    throw(UnsupportedError("IIR filtfilt is a serial recursion; only FIR coefficients run on the device"))

# ------------------------------------------------------------------------------------------------ multitaper
# dpss(n, nw, ntapers)   windows.jl:668-726: the Slepian tapers as eigenvectors of the symmetric tridiagonal matrix (host, like every window)
function dpss(n::Integer, nw::Real, ntapers::Integer=ceil(Int, 2 * nw) - 1)
    0 < ntapers <= n || throw(DomainError(ntapers, "ntapers must be in the interval (0, n]"))
    0 <= nw < n / 2 || throw(DomainError(nw, "nw must be in the interval [0, n/2)"))
    dv = [cospi(2 * nw / n) * ((n - 1) / 2 - i)^2 for i in 0:n-1]
    ev = [0.5 * (k * n - k^2) for k in 1:n-1]
    v = eigen(SymTridiagonal(dv, ev), n-ntapers+1:n).vectors[:, end:-1:1]
    for c in 1:ntapers       # sign conventions :696-707: symmetric tapers have a positive mean, skew ones a positive first lobe
        col = view(v, :, c)
        flip = isodd(c) ? sum(col) < 0 : col[findfirst(x -> abs(x) > 1e-12, col)] < 0
        flip && (col .*= -1)
    end
    v
end

# MTConfig{T}(n_samples; fs, nfft, window, nw, ntapers, taper_weights, onesided)   multitaper.jl:5-49, :112-135
mutable struct MTConfig{T}
    h::Ptr{Cvoid}
    n_samples::Int; fs::Float64; nfft::Int; ntapers::Int
    freq::AbstractVector; window::Matrix{Float64}; onesided::Bool; r::Vector{Float64}; nout::Int
end
function MTConfig{T}(n_samples::Integer; fs::Real=1, nfft::Integer=nextpow(2, n_samples), window::Union{Nothing,AbstractMatrix}=nothing, nw::Real=4,
                     ntapers::Integer=2 * nw - 1, taper_weights::AbstractVector=fill(1 / ntapers, ntapers), onesided::Bool=T <: Real,
                     engine=ENGINE_AUTO) where {T}
    onesided && T <: Complex && throw(ArgumentError("cannot compute one-sided FFT of a complex signal"))     # :115-117
    n_samples > 0 || throw(ArgumentError("`n_samples` must be positive"))                                   # :118
    nfft >= n_samples || throw(ArgumentError("Must have `nfft >= n_samples`"))                              # :119
    ntapers > 0 || throw(ArgumentError("`ntapers` must be positive"))                                       # :23
    fs > 0 || throw(ArgumentError("`fs` must be positive"))                                                 # :24
    if window === nothing
        r = fs ./ taper_weights                                                                              # :127
        window = dpss(n_samples, nw, ntapers)                                                                # :128
    else
        r = fs .* vec(sum(abs2, window; dims=1)) ./ taper_weights                                            # :130
    end
    size(window) == (n_samples, ntapers) || throw(DimensionMismatch("Must have `size(window) == (n_samples, ntapers)`"))   # :34-36
    size(r) == (ntapers,) || throw(DimensionMismatch("Must have `size(r) == (ntapers,)`"))                  # :37-39
    w = convert(Matrix{Float64}, window); rv = convert(Vector{Float64}, r)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve w rv check(ccall((:mdsp_mt_plan_create, lib), Cint, (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint, Cint, Cint),
                                  p, n_samples, nfft, pointer(w), ntapers, pointer(rv), onesided, mdtype(fftintype(T)), engine))
    no, nt, eng = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0)
    check(ccall((:mdsp_mt_plan_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Cint}), p[], no, nt, eng))
    c = MTConfig{T}(p[], n_samples, fs, nfft, ntapers, onesided ? rfftfreq(nfft, fs) : fftfreq(nfft, fs), w, onesided, rv, Int(no[]))
    finalizer(x -> ccall((:mdsp_mt_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
    c
end
# multitaper PSDs of the frames of every column of `signal`: (nout, K, nch) on the device   mt_pgram! :225-245, mt_spectrogram! :312-330
function mt_psd(config::MTConfig{T}, signal::AbstractVecOrMat, noverlap::Integer) where {T}
    sd = todevice(signal, fftintype(T))
    len = size(sd, 1); nch = ncolumns(sd)
    K = framecount(len, config.n_samples, noverlap)
    out = DeviceArray{fftabs2type(T)}((config.nout, K, nch))
    K > 0 && check(ccall((:mdsp_mt_psd_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                         config.h, sd.ptr, len, noverlap, nch, len, out.ptr, config.nout, K * config.nout, C_NULL))
    download(out)
end
# mt_pgram(s; ...) / mt_pgram(s, config) / mt_pgram!(output, s, config)   multitaper.jl:177-245
function mt_pgram(s::AbstractVector{T}, config::MTConfig) where {T}
    length(s) == config.n_samples || throw(DimensionMismatch("Expected `signal` to be of length `config.n_samples`; got `length(signal)` = $(length(s)) and `config.n_samples` = $(config.n_samples)"))
    Periodogram(mt_psd(config, s, 0)[:, 1, 1], config.freq)
end
mt_pgram(s::AbstractVector{T}; onesided::Bool=T <: Real, nfft::Int=nextfastfft(length(s)), fs::Real=1, nw::Real=4,
         ntapers::Int=ceil(Int, 2nw) - 1, window::Union{Nothing,AbstractMatrix}=nothing) where {T<:Number} =
    mt_pgram(s, MTConfig{fftintype(T)}(length(s); fs, nfft, window, nw, ntapers, onesided))
function mt_pgram!(output::AbstractVector, s::AbstractVector, config::MTConfig)
    length(output) == length(config.freq) || throw(DimensionMismatch("Expected `output` to be of length `length(config.freq)`; got `length(output)` = $(length(output)) and `length(config.freq)` = $(length(config.freq))"))
    copyto!(output, power(mt_pgram(s, config)))
    Periodogram(output, config.freq)
end
# MTSpectrogramConfig / mt_spectrogram / mt_spectrogram!   multitaper.jl:248-392
struct MTSpectrogramConfig{T}
    n_samples::Int
    n_overlap_samples::Int
    time::StepRangeLen{Float64,Base.TwicePrecision{Float64},Base.TwicePrecision{Float64}}
    mt_config::MTConfig{T}
end
function MTSpectrogramConfig(n_samples::Int, mt_config::MTConfig{T}, n_overlap_samples::Int) where {T}
    spw = mt_config.n_samples
    spw > n_overlap_samples || throw(ArgumentError("Need `samples_per_window > n_overlap_samples`; got `samples_per_window` = $spw and `n_overlap_samples` = $n_overlap_samples."))   # :264-266
    hop = spw - n_overlap_samples
    len = n_samples < spw ? 0 : div(n_samples - spw, hop) + 1
    MTSpectrogramConfig{T}(n_samples, n_overlap_samples, (spw / 2 : hop : (len - 1) * hop + spw / 2) / mt_config.fs, mt_config)   # :270
end
MTSpectrogramConfig{T}(n_samples::Int, samples_per_window::Int, n_overlap_samples::Int; fs::Real=1, kwargs...) where {T} =
    MTSpectrogramConfig(n_samples, MTConfig{T}(samples_per_window; fs, kwargs...), n_overlap_samples)
function mt_spectrogram(signal::AbstractVector, config::MTSpectrogramConfig)
    length(signal) == config.n_samples || throw(DimensionMismatch("Expected `signal` to be of length `config.n_samples`; got `length(signal)` = $(length(signal)) and `config.n_samples` = $(config.n_samples)"))
    Spectrogram(mt_psd(config.mt_config, signal, config.n_overlap_samples)[:, :, 1], config.mt_config.freq, config.time)
end
mt_spectrogram(signal::AbstractVector, mt_config::MTConfig, n_overlap::Int=mt_config.n_samples >> 1) =
    mt_spectrogram(signal, MTSpectrogramConfig(length(signal), mt_config, n_overlap))
mt_spectrogram(signal::AbstractVector{T}, n::Int=length(signal) >> 3, n_overlap::Int=n >> 1; fs::Real=1, onesided::Bool=T <: Real, kwargs...) where {T} =
    mt_spectrogram(signal, MTSpectrogramConfig{fftintype(T)}(length(signal), n, n_overlap; fs, onesided, kwargs...))
function mt_spectrogram!(destination::AbstractMatrix, signal::AbstractVector, config::MTSpectrogramConfig)
    size(destination) == (length(config.mt_config.freq), length(config.time)) ||
        throw(DimensionMismatch("Expected `destination` to be of size `(length(config.mt_config.freq), length(config.time))`"))   # :314-317
    copyto!(destination, power(mt_spectrogram(signal, config)))
    Spectrogram(destination, config.mt_config.freq, config.time)
end
# cross power spectra and coherence   multitaper.jl:409-817 (signal: n_channels x n_samples, like the reference)
struct CrossPowerSpectra{T,F,A<:AbstractArray{T,3}} <: TFR{T}
    power::A
    freq::F
end
struct Coherence{T,F,A<:AbstractArray{T,3}} <: TFR{T}
    coherence::A
    freq::F
end
coherence(c::Coherence) = c.coherence
struct MTCrossSpectraConfig{T}
    n_channels::Int
    mt_config::MTConfig{T}
    demean::Bool
    freq_inds::Vector{Int64}          # 1-based indices into mt_config.freq
    freq::AbstractVector
end
function MTCrossSpectraConfig(n_channels::Int, mt_config::MTConfig{T}; demean::Bool=false, freq_range=nothing) where {T}
    (T <: Real && mt_config.onesided) || throw(ArgumentError("Only real data is supported (with the default choice of `onesided=true`) for this operation."))   # :417-422
    inds = freq_range === nothing ? collect(Int64, 1:length(mt_config.freq)) :
           Int64[i for i in eachindex(mt_config.freq) if freq_range[1] < mt_config.freq[i] < freq_range[end]]   # :503
    MTCrossSpectraConfig{T}(n_channels, mt_config, demean, inds, mt_config.freq[inds])
end
MTCrossSpectraConfig{T}(n_channels::Int, n_samples::Int; fs::Real=1, demean::Bool=false, freq_range=nothing, kwargs...) where {T} =
    MTCrossSpectraConfig(n_channels, MTConfig{T}(n_samples; fs, kwargs...); demean, freq_range)
function cross_spectra_device(signal::AbstractMatrix, config::MTCrossSpectraConfig{T}) where {T}
    mc = config.mt_config
    size(signal) == (config.n_channels, mc.n_samples) ||
        throw(DimensionMismatch("Size of `signal` does not match `(config.n_channels, config.mt_config.n_samples)`; got `size(signal)`=$(size(signal))"))   # :557-560
    nch = config.n_channels
    sd = todevice(permutedims(signal), fftintype(T))                     # (n_samples, n_channels): one channel per column
    C = fftouttype(fftintype(T))
    xmt = DeviceArray{C}((mc.nout, mc.ntapers, nch))
    check(ccall((:mdsp_mt_spectra_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                mc.h, sd.ptr, nch, mc.n_samples, config.demean, xmt.ptr, C_NULL))
    out = DeviceArray{C}((nch, nch, length(config.freq_inds)))
    fi = config.freq_inds .- 1
    GC.@preserve fi check(ccall((:mdsp_mt_cross_spectra, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                                mc.h, xmt.ptr, nch, pointer(fi), length(fi), out.ptr, C_NULL))
    out
end
mt_cross_power_spectra(signal::AbstractMatrix, config::MTCrossSpectraConfig) = CrossPowerSpectra(download(cross_spectra_device(signal, config)), config.freq)
mt_cross_power_spectra(signal::AbstractMatrix{T}; fs::Real=1, kwargs...) where {T} =
    mt_cross_power_spectra(signal, MTCrossSpectraConfig{fftintype(T)}(size(signal)...; fs, kwargs...))
function mt_cross_power_spectra!(output::AbstractArray{<:Any,3}, signal::AbstractMatrix, config::MTCrossSpectraConfig)
    size(output) == (config.n_channels, config.n_channels, length(config.freq_inds)) ||
        throw(DimensionMismatch("Size of `output` does not match `(config.n_channels, config.n_channels, length(config.freq_inds))`"))   # :561-564
    copyto!(output, power(mt_cross_power_spectra(signal, config)))
    CrossPowerSpectra(output, config.freq)
end
struct MTCoherenceConfig{T}
    cs_config::MTCrossSpectraConfig{T}
end
MTCoherenceConfig{T}(n_channels::Int, n_samples::Int; kwargs...) where {T} = MTCoherenceConfig{T}(MTCrossSpectraConfig{T}(n_channels, n_samples; kwargs...))
function mt_coherence(signal::AbstractMatrix, config::MTCoherenceConfig{T}) where {T}
    cs = cross_spectra_device(signal, config.cs_config)
    nch, nfi = size(cs, 1), size(cs, 3)
    R = fftabs2type(fftintype(T))
    out = DeviceArray{R}((nch, nch, nfi))
    check(ccall((:mdsp_coherence_from_cs, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}), cs.ptr, nch, nfi, mdtype(R), out.ptr, C_NULL))   # coherence_from_cs!, :704-723
    Coherence(download(out), config.cs_config.freq)
end
mt_coherence(signal::AbstractMatrix{T}; fs::Real=1, freq_range=nothing, demean::Bool=false, kwargs...) where {T} =
    mt_coherence(signal, MTCoherenceConfig{fftintype(T)}(size(signal)...; fs, freq_range, demean, kwargs...))
function mt_coherence!(output::AbstractArray{<:Any,3}, signal::AbstractMatrix, config::MTCoherenceConfig)
    size(output) == (config.cs_config.n_channels, config.cs_config.n_channels, length(config.cs_config.freq)) ||
        throw(DimensionMismatch("Size of `output` does not match `(config.cs_config.n_channels, config.cs_config.n_channels, length(config.cs_config.freq))`"))   # :774-777
    copyto!(output, coherence(mt_coherence(signal, config)))
    Coherence(output, config.cs_config.freq)
end

# ------------------------------------------------------------------------------------------------ yardsticks
# float4 copy / read / fill streams on the device (the bandwidth yardsticks bench.py quotes next to the kernels)
copy_bench!(dst::DeviceArray, src::DeviceArray) =
    check(ccall((:mdsp_copy_bench, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), dst.ptr, src.ptr, length(src) * sizeof(eltype(src)), C_NULL))
copy_bench!(dst::DeviceArray, src::DeviceArray, mode::Integer, wgs_per_cu::Integer) =
    check(ccall((:mdsp_copy_bench_mode, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint, Cint, Ptr{Cvoid}), dst.ptr, src.ptr, length(src) * sizeof(eltype(src)), mode, wgs_per_cu, C_NULL))

end # module
