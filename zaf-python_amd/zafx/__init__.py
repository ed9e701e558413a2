"""zafx -- MI355X-native drop-in for the windowed-transform path of zaf.py.

    import zafx as zaf          # same signatures as zafarrafii/Zaf-Python for this path
    X = zaf.stft(x, w, 1024)    # runs on the GPU through libzafx.so (hand-written HIP)

Drop-in functions (zaf.py signatures, float64 / complex128 results):
    stft, istft, melfilterbank, melspectrogram, mfcc, cqtkernel, cqtspectrogram,
    cqtchromagram, mdct, imdct, dct, dst
Batched extension ((clips, samples) in, float32 / complex64 out):
    stft_batch, istft_batch, mdct_batch, imdct_batch, melspectrogram_batch, mfcc_batch,
    cqtspectrogram_batch, cqtchromagram_batch, dct_batch, dst_batch, mel_mfcc_batch (melspectrogram + mfcc from one set of transforms);
    the same from interleaved int16 / int32 PCM:
    stft_pcm_batch, mdct_pcm_batch, melspectrogram_pcm_batch, mfcc_pcm_batch, cqtspectrogram_pcm_batch, cqtchromagram_pcm_batch
Device-resident API: Plan, DeviceBuffer, Comm, *_plan factories, shard helpers; one process per GPU: launch.Rendezvous,
spawn_ranks (file rendezvous + self-launcher, no torch.distributed).
"""
from ._lib import (CHROMA, CQT, DCT, IMDCT, ISTFT, LAYOUT_FT, LAYOUT_TF, LINEAR, MDCT, MEL, MFCC, STFT, ZafxError, device_count,
                   device_name, library_path)
from .constants import cqtkernel, dct2_rows, dct_matrix, dst_matrix, hamming, kaiser_bessel_derived, melfilterbank, sine
from .core import (Comm, DeviceBuffer, Plan, clear_plan_cache, cqt_plan, cqtchromagram, cqtchromagram_batch, dct, dct_batch, dst,
                   dst_batch, linear_plan, dct_plan, dct_fft_length,
                   cqtspectrogram, cqtspectrogram_batch, imdct, imdct_batch, istft, istft_batch, istft_plan, mdct,
                   mdct_batch, mdct_plan, mel_plan, melspectrogram, melspectrogram_batch, mfcc, mfcc_batch, pcm_to_mono, pinned_empty,
                   get_precision, set_precision, stft, stft_batch, stft_pcm_batch, stft_plan, mdct_pcm_batch, melspectrogram_pcm_batch, mfcc_pcm_batch,
                   cqtspectrogram_pcm_batch, cqtchromagram_pcm_batch, mel_mfcc_batch, mel_mfcc_pcm_batch, mel_mfcc_supported, set_row_padding, get_row_padding)
from .launch import Rendezvous, rank_env, spawn_ranks
from .shard import clip_range, run_sharded, shard_sizes

__version__ = "0.1.0"
