from .base import EncoderModule

__all__ = ['EncoderModule']
