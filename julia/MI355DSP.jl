# MI355DSP.jl -- thin Julia host for libmi355dsp.so (the C ABI of include/mi355dsp.h).
#
# STATUS: written against the header, NOT executed -- the build image has no Julia.  The same C ABI is exercised
# by the Python/ctypes host (dsp.jl_amd/) and its GPU parity tests; this file is the `ccall` twin a DSP.jl
# maintainer would use.  It keeps DSP.jl's names, argument order, defaults, promotion rules and exception types
# for the hot path (filt / fftfilt / conv / periodogram / welch_pgram / spectrogram / stft / FIRFilter / resample),
# and does nothing else: no CUDA.jl / AMDGPU.jl array dispatch, every call goes straight to a hand-written HIP
# kernel through `ccall`.
#
# Host `Array`s go through the library's host-pointer pipelines (mdsp_ols_exec_host / mdsp_welch_exec_host: pinned double buffers,
# H2D || kernel || D2H on two streams; `pin!` page-locks an Array so the DMA engines use it directly); `DeviceArray` keeps data
# resident in HBM between calls.  The function-style calls take their plans from the library's own LRU (mdsp_*_plan_cached), the
# multi-GPU part is `Comm` (RCCL behind the C ABI).  Reference locations are DSP.jl v0.8.5 `src/`.
module MI355DSP

export DeviceArray, upload, download, filt, fftfilt, tdfilt, conv, periodogram, WelchConfig, welch_pgram,
       spectrogram, stft, FIRFilter, resample, reset!, setphase!, timedelay, inputlength, outputlength,
       nextfastfft, optimalfftfiltlength, Comm, welch_channel_mean, welch_reset!, welch_accumulate!, welch_finalize, welch_allreduce!,
       pin!, unpin!

const lib = get(ENV, "MI355DSP_LIB", joinpath(@__DIR__, "..", "dsp.jl_amd", "libmi355dsp.so"))

# ---------------------------------------------------------------------------------------------- status -> exception
const MDSP_F32, MDSP_F64, MDSP_C32, MDSP_C64 = Cint(0), Cint(1), Cint(2), Cint(3)
const ENGINE_AUTO, ENGINE_FUSED, ENGINE_ROCFFT = Cint(0), Cint(1), Cint(2)

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

# element-type rules, util.jl:92-104
fftintype(::Type{T}) where {T<:Union{Float32,Float64,ComplexF32,ComplexF64}} = T
fftintype(::Type{T}) where {T<:Real} = Float64
fftintype(::Type{T}) where {T<:Complex} = ComplexF64
fftouttype(::Type{T}) where {T<:Union{ComplexF32,ComplexF64}} = T
fftouttype(::Type{Float32}) = ComplexF32
fftouttype(::Type{T}) where {T<:Union{Real,Complex}} = ComplexF64
fftabs2type(::Type{T}) where {T<:Union{Float32,ComplexF32}} = Float32
fftabs2type(::Type{T}) where {T<:Union{Real,Complex}} = Float64

init(device::Integer=0) = check(ccall((:mdsp_init, lib), Cint, (Cint,), device))

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
Base.length(a::DeviceArray) = prod(a.dims)
Base.eltype(::DeviceArray{T}) where {T} = T

function upload(x::Array{T,N}) where {T,N}
    d = DeviceArray{T}(size(x))
    check(ccall((:mdsp_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), d.ptr, x, sizeof(x), C_NULL))
    d
end
function download(d::DeviceArray{T,N}) where {T,N}
    x = Array{T,N}(undef, d.dims)
    check(ccall((:mdsp_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}), x, d.ptr, sizeof(x), C_NULL))
    x
end
todevice(x::DeviceArray, ::Type{T}) where {T} = eltype(x) == T ? x : upload(convert(Array{T}, download(x)))
todevice(x::AbstractArray, ::Type{T}) where {T} = upload(convert(Array{T}, x))
back(y::DeviceArray, like::DeviceArray) = y
back(y::DeviceArray, like) = download(y)

# ---------------------------------------------------------------------------------------------- index arithmetic
nextfastfft(n::Integer) = Int(ccall((:mdsp_nextfastfft, lib), Int64, (Int64,), n))                       # util.jl:134
optimalfftfiltlength(nb, nx) = Int(ccall((:mdsp_optimal_fft_len, lib), Int64, (Int64, Int64), nb, nx))  # dspbase.jl:268
framecount(len, n, noverlap) = Int(ccall((:mdsp_frame_count, lib), Int64, (Int64, Int64, Int64), len, n, noverlap))
outputlength(inlen::Integer, ratio::Union{Integer,Rational}, ϕ::Integer) =                                 # stream_filt.jl:317
    Int(ccall((:mdsp_outputlength, lib), Int64, (Int64, Int64, Int64, Int64), inlen, numerator(ratio), denominator(ratio), ϕ))
inputlength(outlen::Integer, ratio::Union{Integer,Rational}, ϕ::Integer, r::RoundingMode=RoundDown) =      # stream_filt.jl:358
    Int(ccall((:mdsp_inputlength, lib), Int64, (Int64, Int64, Int64, Int64, Cint), outlen, numerator(ratio), denominator(ratio), ϕ,
              (r == RoundUp || r == RoundFromZero) ? 1 : 0))

const SMALL_FILT_CUTOFF = 66   # dspbase.jl:3

# ---------------------------------------------------------------------------------------------- overlap-save filt / conv
mutable struct OlsPlan
    h::Ptr{Cvoid}
    function OlsPlan(taps::Vector{T}, nfft::Integer, nx::Integer, mode::Integer, engine=ENGINE_AUTO) where {T}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mdsp_ols_plan_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Cint),
                    p, taps, length(taps), nfft, nx, mdtype(T), mode, engine))
        o = new(p[])
        finalizer(x -> ccall((:mdsp_ols_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.h), o)
        o
    end
end

function olsexec(plan::OlsPlan, x::DeviceArray{T}, nout::Integer) where {T}
    nx = size(x, 1); ncols = length(x) ÷ max(nx, 1)
    y = DeviceArray{T}((nout, size(x)[2:end]...))
    check(ccall((:mdsp_ols_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                plan.h, x.ptr, nx, ncols, nx, y.ptr, nout, nout, C_NULL))
    y
end

# The function-style entry points build a plan per call in the reference (cheap FFTW plans); here the plan comes from the LIBRARY's LRU
# (mdsp_ols_plan_cached: keyed by device, thread, stream and the contents of the taps) -- ~45 us per call instead of ~1 ms.  The handle is
# borrowed: no finalizer, never destroyed from here.
function cached_ols_plan(taps::Vector{T}, nfft::Integer, nx::Integer, mode::Integer, engine=ENGINE_AUTO) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mdsp_ols_plan_cached, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Cint, Ptr{Cvoid}),
                p, taps, length(taps), nfft, nx, mdtype(T), mode, engine, C_NULL))
    p[]
end

# page-lock an existing Array (hipHostRegister) so that the host pipelines DMA straight from / into it
pin!(x::Array) = (check(ccall((:mdsp_host_register, lib), Cint, (Ptr{Cvoid}, Csize_t), x, sizeof(x))); x)
unpin!(x::Array) = (check(ccall((:mdsp_host_unregister, lib), Cint, (Ptr{Cvoid},), x)); x)
const HOST_PINNED = Cint(1)

# fftfilt(b, x[, nfft])   Filters/filt.jl:458-461, _fftfilt! :479-521
# Device arrays: one launch sequence on resident data.  Host Arrays: the chunked H2D || kernel || D2H pipeline (same block grid, so
# both return bit-identical results).
function fftfilt(b::AbstractVector{H}, x::DeviceArray{T}, nfft::Integer=optimalfftfiltlength(length(b), length(x))) where {H<:Real,T<:Real}
    W = fftintype(promote_type(H, T))
    xd = todevice(x, W)
    nx = size(xd, 1); ncols = length(xd) ÷ max(nx, 1)
    plan = cached_ols_plan(convert(Vector{W}, b), nfft, nx, 0)
    y = DeviceArray{W}(size(xd))
    check(ccall((:mdsp_ols_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                plan, xd.ptr, nx, ncols, nx, y.ptr, nx, nx, C_NULL))
    y
end
function fftfilt(b::AbstractVector{H}, x::AbstractArray{T}, nfft::Integer=optimalfftfiltlength(length(b), length(x));
                 pinned::Bool=false) where {H<:Real,T<:Real}
    W = fftintype(promote_type(H, T))
    xh = convert(Array{W}, x)                                  # column-major (n, cols...): exactly the layout the C ABI takes
    nx = size(xh, 1); ncols = length(xh) ÷ max(nx, 1)
    plan = cached_ols_plan(convert(Vector{W}, b), nfft, nx, 0)
    y = similar(xh)
    check(ccall((:mdsp_ols_exec_host, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Cint),
                plan, xh, nx, ncols, nx, y, nx, nx, pinned ? HOST_PINNED : Cint(0)))
    y
end

# filt(b, a::Number, x) / tdfilt   dspbase.jl:14-66 (FIR only on the device)
function filt(b::AbstractVector, a::Number, x::Union{AbstractArray{T},DeviceArray{T}}) where {T}
    isempty(b) && throw(ArgumentError("filter vector b must be non-empty"))
    a == 0 && throw(ArgumentError("filter vector a[1] must be nonzero"))
    W = fftintype(promote_type(eltype(b), typeof(a), T))
    R = real(W)
    taps = convert(Vector{R}, a == 1 ? b : b ./ a)
    xd = todevice(x, W)
    y = DeviceArray{W}(size(xd))
    nx = size(xd, 1); ncols = length(xd) ÷ max(nx, 1)
    check(ccall((:mdsp_tdfir_exec, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                taps, length(taps), mdtype(W), xd.ptr, nx, ncols, nx, y.ptr, nx, C_NULL))
    back(y, x)
end
tdfilt(h::AbstractVector{H}, x) where {H} = filt(h, one(H), x)                      # filt.jl:431

# filt(b, x): FFT path for real taps longer than SMALL_FILT_CUTOFF, time domain otherwise   filt.jl:525-555
function filt(b::AbstractVector{<:Real}, x::Union{AbstractArray{<:Real},DeviceArray{<:Real}})
    length(b) > SMALL_FILT_CUTOFF ? fftfilt(b, x, optimalfftfiltlength(length(b), size(x, 1))) : tdfilt(b, x)
end

# conv(u, v; algorithm)   dspbase.jl:709-792 (vectors; :direct for small / integer inputs stays on the CPU)
function conv(u::AbstractVector{Tu}, v::AbstractVector{Tv}; algorithm=:auto) where {Tu<:Number,Tv<:Number}
    T = promote_type(Tu, Tv)
    W = fftintype(T)
    nu, nv = length(u), length(v)
    alg = algorithm
    alg === :auto && (alg = T <: Union{Float32,Float64,ComplexF32,ComplexF64} ? :fast : :direct)
    alg === :fast && (alg = nu * nv < 2^16 ? :direct : :fft)
    (alg === :direct || nu == 0 || nv == 0) && throw(UnsupportedError("direct convolution: use DSP.conv on the CPU"))
    big, small = nu >= nv ? (u, v) : (v, u)
    os = optimalfftfiltlength(length(small), length(big))
    alg === :fft && (alg = os < nu + nv - 1 ? :fft_overlapsave : :fft_simple)
    alg in (:fft_overlapsave, :fft_simple) ||
        throw(ArgumentError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave"))
    nfft = alg === :fft_simple ? nextfastfft(nu + nv - 1) : os
    bigd = todevice(big, W)
    plan = cached_ols_plan(convert(Vector{W}, small), nfft, length(big), 1)
    y = DeviceArray{W}((nu + nv - 1,))
    check(ccall((:mdsp_ols_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                plan, bigd.ptr, length(big), 1, length(big), y.ptr, nu + nv - 1, nu + nv - 1, C_NULL))
    download(y)
end

# conv(u, v; algorithm) for arrays   dspbase.jl:709-792: _conv_kern_fft! (:611-644) / _conv_td! (:646-660) on the device.
# Julia arrays are column-major, which is the layout the C ABI takes: sizes are passed as they are.
function conv(u::AbstractArray{Tu,N}, v::AbstractArray{Tv,N}; algorithm=:auto) where {Tu<:Number,Tv<:Number,N}
    T = promote_type(Tu, Tv)
    W = T <: Union{Float32,Float64,ComplexF32,ComplexF64} ? T : (T <: Complex ? ComplexF64 : Float64)
    so = size(u) .+ size(v) .- 1
    (isempty(u) || isempty(v)) && return zeros(T, max.(so, 0))
    alg = algorithm
    alg === :auto && (alg = T === W ? :fast : :direct)
    alg === :fast && (alg = length(u) * length(v) < 2^16 ? :direct : :fft)
    alg in (:direct, :fft, :fft_simple, :fft_overlapsave) ||
        throw(ArgumentError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave"))
    ud, vd, out = todevice(u, W), todevice(v, W), DeviceArray{W}(so)
    entry = alg === :direct ? :mdsp_convnd_direct : :mdsp_convnd_fft
    su, sv = collect(Int64, size(u)), collect(Int64, size(v))
    if entry === :mdsp_convnd_direct
        check(ccall((:mdsp_convnd_direct, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                    ud.ptr, su, vd.ptr, sv, N, mdtype(W), out.ptr, C_NULL))
    else
        check(ccall((:mdsp_convnd_fft, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Int64}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                    ud.ptr, su, vd.ptr, sv, N, mdtype(W), out.ptr, C_NULL))
    end
    res = download(out)
    T === W ? res : round.(T, res)                     # integer eltypes: exact in Float64 below 2^53
end
# rank promotion with trailing singleton dimensions   dspbase.jl:784-792
function conv(A::AbstractArray{<:Number,M}, B::AbstractArray{<:Number,N}; kwargs...) where {M,N}
    M < N ? conv(reshape(A, size(A)..., ntuple(_ -> 1, N - M)...), B; kwargs...) :
            conv(A, reshape(B, size(B)..., ntuple(_ -> 1, M - N)...); kwargs...)
end

# ---------------------------------------------------------------------------------------------- periodograms
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
winptr(::Nothing) = Ptr{Float64}(C_NULL)
winptr(w::Vector{Float64}) = pointer(w)

struct WelchConfig                                                                        # periodograms.jl:516-587
    h::Base.RefValue{Ptr{Cvoid}}
    nsamples::Int; noverlap::Int; onesided::Bool; nfft::Int; fs::Float64
    freq::AbstractVector; window; r::Float64; intype::DataType; nout::Int
end
function WelchConfig(nsamples, ::Type{T}; n::Int=nsamples >> 3, noverlap::Int=n >> 1, onesided::Bool=T <: Real,
                     nfft::Int=nextfastfft(n), fs::Real=1, window=nothing, engine=ENGINE_AUTO) where {T}
    onesided && T <: Complex && throw(ArgumentError("cannot compute one-sided FFT of a complex signal"))
    nfft >= n || throw(DomainError((; nfft, n), "nfft must be >= n"))
    win, norm2 = compute_window(window, n)
    r = fs * norm2
    S = fftintype(T)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve win check(ccall((:mdsp_welch_plan_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Float64}, Cdouble, Cint, Cint, Cint),
        p, n, noverlap, nfft, winptr(win), r, onesided, mdtype(S), engine))
    nout = onesided ? (nfft >> 1) + 1 : nfft
    freq = onesided ? (0:nfft>>1) .* (fs / nfft) : vcat(0:(nfft-1)>>1, -(nfft >> 1):-1) .* (fs / nfft)
    cfg = WelchConfig(p, n, noverlap, onesided, nfft, fs, freq, win, r, S, nout)
    finalizer(x -> ccall((:mdsp_welch_plan_destroy, lib), Cint, (Ptr{Cvoid},), x[]), p)
    cfg
end
WelchConfig(data::AbstractArray; kw...) = WelchConfig(size(data, ndims(data)), eltype(data); kw...)

# welch_pgram(s, config) -> (power, freq)   periodograms.jl:702-705, :746-759.  Columns of a matrix are channels.
function welch_pgram(s::Union{AbstractVecOrMat{T},DeviceArray{T}}, config::WelchConfig) where {T<:Number}
    fftintype(T) == config.intype ||
        throw(ArgumentError("float(eltype(s)) = $T doesn't match the eltype of the input buffer: $(config.intype)."))
    sd = todevice(s, config.intype)
    len = size(sd, 1); nch = length(sd) ÷ max(len, 1)
    out = DeviceArray{fftabs2type(config.intype)}(nch == 1 && length(size(sd)) == 1 ? (config.nout,) : (config.nout, nch))
    check(ccall((:mdsp_welch_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                config.h[], sd.ptr, len, nch, len, out.ptr, config.nout, C_NULL))
    (power = back(out, s), freq = config.freq)
end
welch_pgram(s::AbstractVector, n::Int=length(s) >> 3, noverlap::Int=n >> 1; kw...) =
    welch_pgram(s, WelchConfig(s; n, noverlap, kw...))

# Host Arrays: mdsp_welch_exec_host (time chunks of whole frames through pinned double buffers; Float64 sums accumulate in the plan)
function welch_pgram(s::Array{T}, config::WelchConfig; pinned::Bool=false) where {T<:Union{Float32,Float64,ComplexF32,ComplexF64}}
    T == config.intype || throw(ArgumentError("float(eltype(s)) = $T doesn't match the eltype of the input buffer: $(config.intype)."))
    len = size(s, 1); nch = length(s) ÷ max(len, 1)
    out = Array{fftabs2type(T)}(undef, ndims(s) == 1 ? (config.nout,) : (config.nout, nch))
    check(ccall((:mdsp_welch_exec_host, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Cint),
                config.h[], s, len, nch, len, out, config.nout, pinned ? HOST_PINNED : Cint(0)))
    (power = out, freq = config.freq)
end

# Streaming form (mdsp_welch_exec IS reset + accumulate + finalize): a stream handed over slice by slice -- each slice whole frames,
# consecutive slices overlapping by n - hop samples -- or by several ranks (welch_allreduce!).   periodograms.jl:746-759
welch_reset!(c::WelchConfig) = (check(ccall((:mdsp_welch_reset, lib), Cint, (Ptr{Cvoid},), c.h[])); c)
function welch_accumulate!(c::WelchConfig, s::DeviceArray)
    len = size(s, 1); nch = length(s) ÷ max(len, 1)
    check(ccall((:mdsp_welch_accumulate, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}), c.h[], s.ptr, len, nch, len, C_NULL))
    c
end
function welch_finalize(c::WelchConfig, nch::Integer=1; frames_total::Integer=0)
    out = DeviceArray{fftabs2type(c.intype)}((c.nout, nch))
    check(ccall((:mdsp_welch_finalize, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}), c.h[], frames_total, out.ptr, c.nout, C_NULL))
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
    check(ccall((:mdsp_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    id
end
function Comm(id::Vector{UInt8}, rank::Integer, nranks::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mdsp_comm_init_rank, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{UInt8}, Cint, Cint), p, id, rank, nranks))
    c = Comm(p[], rank, nranks)
    finalizer(x -> ccall((:mdsp_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
    c
end
channel_shard(nch::Integer, rank::Integer, nranks::Integer) = (per = cld(nch, nranks); (min(rank * per, nch) + 1):min((rank + 1) * per, nch))

# mean over ALL channels (across ranks) of the per-channel Welch PSDs: local sum on the device -> ncclAllReduce(sum) of nout values ->
# 1/nch_total, one C-ABI call.  `s`: this rank's channels as columns.
function welch_channel_mean(s::DeviceArray{T}, config::WelchConfig, nch_total::Integer, comm::Union{Comm,Nothing}=nothing) where {T}
    len = size(s, 1); nloc = length(s) ÷ max(len, 1)
    R = fftabs2type(config.intype)
    psd = DeviceArray{R}((config.nout, max(nloc, 1)))
    nloc > 0 && check(ccall((:mdsp_welch_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
                            config.h[], s.ptr, len, nloc, len, psd.ptr, config.nout, C_NULL))
    mean = DeviceArray{R}((config.nout,))
    check(ccall((:mdsp_welch_mean_allreduce, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                config.h[], psd.ptr, nloc, config.nout, nch_total, mean.ptr, comm === nothing ? C_NULL : comm.h, C_NULL))
    mean
end
# one stream split along time over ranks: after welch_accumulate! on every rank's slice, sum the Float64 accumulators and frame counts
welch_allreduce!(c::WelchConfig, comm::Comm) =
    (check(ccall((:mdsp_welch_allreduce, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), c.h[], comm.h, C_NULL)); c)

# stft / spectrogram / periodogram   periodograms.jl:872-897, :828-837, :393-417
function stft(s::Union{AbstractVecOrMat{T},DeviceArray{T}}, n::Int=size(s, 1) >> 3, noverlap::Int=n >> 1, psdonly::Bool=false;
              onesided::Bool=T <: Real, nfft::Int=nextfastfft(n), fs::Real=1, window=nothing, engine=ENGINE_AUTO) where {T}
    onesided && T <: Complex && throw(ArgumentError("cannot compute one-sided FFT of a complex signal"))
    win, norm2 = compute_window(window, n)
    (0 ≤ noverlap < n) || throw(DomainError((; noverlap, n), "noverlap must be between zero and n"))
    nfft >= n || throw(DomainError((; nfft, n), "nfft must be >= n"))
    S = fftintype(T)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve win check(ccall((:mdsp_stft_plan_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Float64}, Cdouble, Cint, Cint, Cint, Cint),
        p, n, noverlap, nfft, winptr(win), fs * norm2, onesided, psdonly, mdtype(S), engine))
    sd = todevice(s, S)
    len = size(sd, 1); nch = length(sd) ÷ max(len, 1)
    nout = onesided ? (nfft >> 1) + 1 : nfft
    k = framecount(len, n, noverlap)
    out = DeviceArray{psdonly ? fftabs2type(S) : fftouttype(S)}(nch == 1 ? (nout, k) : (nout, k, nch))
    try
        k > 0 && check(ccall((:mdsp_stft_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                             p[], sd.ptr, len, nch, len, out.ptr, nout, nout * k, C_NULL))
    finally
        ccall((:mdsp_stft_plan_destroy, lib), Cint, (Ptr{Cvoid},), p[])
    end
    back(out, s)
end
function spectrogram(s, n::Int=size(s, 1) >> 3, noverlap::Int=n >> 1; onesided::Bool=eltype(s) <: Real,
                     nfft::Int=nextfastfft(n), fs::Real=1, window=nothing)
    out = stft(s, n, noverlap, true; onesided, nfft, fs, window)
    (power = out, freq = onesided ? (0:nfft>>1) .* (fs / nfft) : vcat(0:(nfft-1)>>1, -(nfft >> 1):-1) .* (fs / nfft),
     time = (n / 2 : n - noverlap : (size(out, 2) - 1) * (n - noverlap) + n / 2) / fs)
end
function periodogram(s::AbstractVector{T}; onesided::Bool=T <: Real, nfft::Int=nextfastfft(length(s)), fs::Real=1, window=nothing) where {T}
    nfft >= length(s) || throw(DomainError((; nfft, n=length(s)), "nfft must be >= n = length(s)"))
    p = stft(s, length(s), 0, true; onesided, nfft, fs, window)
    (power = vec(p), freq = onesided ? (0:nfft>>1) .* (fs / nfft) : vcat(0:(nfft-1)>>1, -(nfft >> 1):-1) .* (fs / nfft))
end

# ---------------------------------------------------------------------------------------------- FIRFilter / resample
mutable struct FIRFilter                                                                 # stream_filt.jl:137-178
    h::Ptr{Cvoid}
    taps::Vector
    ratio::Rational{Int}
    xtype::DataType
    nch::Int
end
function FIRFilter(taps::Vector{Th}, ratio::Union{Integer,Rational}=1; xtype::DataType=Th, nch::Integer=1) where {Th<:Union{Float32,Float64}}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    r = convert(Rational{Int}, ratio)
    check(ccall((:mdsp_fir_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Int64),
                p, taps, length(taps), numerator(r), denominator(r), mdtype(Th), mdtype(xtype), nch))
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
# which device kernel a chunk of `n` samples would take: 0 generic, 1 register-tap, 2 matrix-core (diagnostics; results do not depend on it)
function kernel_path(f::FIRFilter, n::Integer)
    p = Ref{Cint}(-1)
    check(ccall((:mdsp_fir_kernel_path, lib), Cint, (Ptr{Cvoid}, Int64, Ref{Cint}), f.h, n, p)); Int(p[])
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
function filt(f::FIRFilter, x::Union{AbstractVecOrMat,DeviceArray})
    xd = todevice(x, f.xtype)
    xlen = size(xd, 1)
    ycap = max(outputlength(f, xlen), 0)
    Ty = promote_type(eltype(f.taps), f.xtype)
    y = DeviceArray{Ty}(f.nch == 1 ? (ycap,) : (ycap, f.nch))
    nw = Ref{Int64}(0)
    check(ccall((:mdsp_fir_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ref{Int64}, Ptr{Cvoid}),
                f.h, xd.ptr, xlen, xlen, y.ptr, ycap, max(ycap, 1), nw, C_NULL))
    nw[] == ycap || throw(AssertionError("Length of resampled output different from expectation."))
    back(y, x)
end

# resample(x, rate, h)   stream_filt.jl:688-725 (vector) / :747-775 (array, dims = 1: columns are channels)
function resample(x::AbstractVecOrMat{T}, rate::Union{Integer,Rational}, h::Vector) where {T}
    S = fftintype(T)
    nch = size(x, 2)
    f = FIRFilter(convert(Vector{real(promote_type(eltype(h), S)) == Float32 ? Float32 : Float64}, h), rate; xtype=S, nch)
    rate == 1 || setphase!(f, timedelay(f))                              # undelay!, :706-714
    outLen = ceil(Int, size(x, 1) * rate)
    npad = inputlength(f, outLen, RoundUp)
    xp = zeros(S, npad, nch); xp[1:size(x, 1), :] .= x                   # _zeropad, :699
    y = filt(f, ndims(x) == 1 ? vec(xp) : xp)
    size(y, 1) >= outLen || throw(AssertionError("Resample output shorter than expected."))
    ndims(x) == 1 ? y[1:outLen] : y[1:outLen, :]
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
    check(ccall((:mdsp_firarb_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64, Cdouble, Int64, Cint, Cint, Int64),
                p, taps, length(taps), Float64(rate), Nϕ, mdtype(Th), mdtype(xtype), nch))
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
    Ty = promote_type(eltype(f.taps), f.xtype)
    y = DeviceArray{Ty}(f.nch == 1 ? (ycap,) : (ycap, f.nch))
    nw = Ref{Int64}(0)
    check(ccall((:mdsp_firarb_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ref{Int64}, Ptr{Cvoid}),
                f.h, xd.ptr, xlen, xlen, y.ptr, ycap, ycap, nw, C_NULL))
    yh = back(y, x)
    ndims(yh) == 1 ? yh[1:nw[]] : yh[1:nw[], :]
end
# resample(x, rate::AbstractFloat, h, Nϕ = 32)   stream_filt.jl:692-694, :752-755
function resample(x::AbstractVecOrMat{T}, rate::AbstractFloat, h::Vector, Nϕ::Integer=32) where {T}
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

# ------------------------------------------------------------------------------------------------ multitaper
# MTConfig{T}(n_samples; ...) keeps DSP.jl's own constructor (host: dpss, r, freq); the device plan mirrors it.
mutable struct MTPlan
    h::Ptr{Cvoid}
    nout::Int
end
function MTPlan(::Type{T}, window::Matrix{Float64}, r::Vector{Float64}, nfft::Integer, onesided::Bool) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mdsp_mt_plan_create, lib), Cint, (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Cint, Cint, Cint),
                p, size(window, 1), nfft, window, size(window, 2), r, onesided, mdtype(T), 0))
    pl = MTPlan(p[], onesided ? nfft >> 1 + 1 : nfft)
    finalizer(x -> ccall((:mdsp_mt_plan_destroy, lib), Cint, (Ptr{Cvoid},), x.h), pl)
    pl
end
# mt_pgram!(output, signal, config) / mt_spectrogram!(destination, signal, config)   multitaper.jl:225-245, :312-330
function mt_psd!(out::DeviceArray, pl::MTPlan, s::DeviceArray, noverlap::Integer=0)
    check(ccall((:mdsp_mt_psd_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}),
                pl.h, s.ptr, size(s, 1), noverlap, size(s, 2), size(s, 1), out.ptr, pl.nout, size(out, 1) * size(out, 2), C_NULL))
    out
end
# mt_cross_power_spectra!(output, signal, config)   multitaper.jl:551-585 (signal: samples x channels on the device)
function mt_cross!(out::DeviceArray, xmt::DeviceArray, pl::MTPlan, s::DeviceArray, freq_inds::Vector{Int64}, demean::Bool)
    nch = size(s, 2)
    check(ccall((:mdsp_mt_spectra_exec, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}),
                pl.h, s.ptr, nch, size(s, 1), demean, xmt.ptr, C_NULL))
    check(ccall((:mdsp_mt_cross_spectra, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                pl.h, xmt.ptr, nch, freq_inds .- 1, length(freq_inds), out.ptr, C_NULL))
    out
end

end # module
