"""zonos/utilities/utils.py:6-29,72-97 equivalents."""
import torch


def find_multiple(n: int, k: int) -> int:
    return n if k == 0 or n % k == 0 else n + k - (n % k)


def get_device() -> torch.device:
    return torch.device(torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


DEFAULT_DEVICE = get_device()


# ------------------------------------------------------------------ wire formats on the output side of the path
def save_wav_pcm16(path, wav: torch.Tensor, sample_rate: int = 44100) -> str:
    """Signed 16-bit PCM WAV file (what utilities/cache_utils.py:380-390 writes through torchaudio.save(encoding="PCM_S")
    of a float waveform).  `wav` is either the int16 `[T, channels]` tensor of `DACAutoencoder.decode_to_int16`
    (autoencoder.py:142-170) or a float `[channels, T]` / `[T]` waveform in [-1, 1], converted like `decode_to_int16`
    (clamp, x 32767, truncate toward zero)."""
    import wave

    if wav.dtype == torch.int16:
        pcm = wav if wav.dim() == 2 else wav.unsqueeze(1)                       # [T, C]
    else:
        f = wav.detach().to(torch.float32).cpu()
        f = f.unsqueeze(0) if f.dim() == 1 else f.reshape(-1, f.shape[-1])       # [C, T]
        pcm = (f.clamp(-1.0, 1.0) * 32767.0).to(torch.int16).t()
    pcm = pcm.cpu().contiguous()
    with wave.open(str(path), "wb") as w:
        w.setnchannels(int(pcm.shape[1]))
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(pcm.numpy().astype("<i2").tobytes())
    return str(path)


def save_tensor_cache(path, tensor: torch.Tensor) -> str:
    """A speaker-embedding / prefix-code cache file as the reference keeps them (utilities/cache_utils.py:322-338:
    `torch.save(tensor, <key>.pt)`), readable by the reference's `load_from_disk` and by `load_tensor_cache`."""
    torch.save(tensor.detach().cpu(), str(path))
    return str(path)


def load_tensor_cache(path, device=None):
    """utilities/cache_utils.py:340-362: `torch.load(file, map_location=device, weights_only=True)`; None when the file
    does not exist.  weights_only=True executes nothing from the file, so caches written by the reference load as they are."""
    import os

    if not os.path.exists(str(path)):
        return None
    return torch.load(str(path), map_location=device, weights_only=True)
