"""zonos/utilities/utils.py:6-29,72-97 equivalents."""
import torch


def find_multiple(n: int, k: int) -> int:
    return n if k == 0 or n % k == 0 else n + k - (n % k)


def get_device() -> torch.device:
    return torch.device(torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


DEFAULT_DEVICE = get_device()
