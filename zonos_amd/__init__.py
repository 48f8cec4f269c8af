"""zonos_amd — MI355X-native (gfx950) implementation of the Zonos TTS hot path: the autoregressive DAC-token
decode loop and DAC decode, behind the reference's `Zonos.generate()` / `DACAutoencoder.decode()` surface."""
__version__ = "0.1.0"
