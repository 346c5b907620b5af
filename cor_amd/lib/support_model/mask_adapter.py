"""Mask pooling (RRE) parameter tree (names of lib/support_model/mask_adapter.py:28-50,83-94,97-142,182-208,226-241)."""
import torch
from torch import nn


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format = eps, data_format


class MaskedPooling(nn.Module):
    pass


class ChannelReduction(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Conv2d(in_channel, out_channel, 1)
        self.norm = LayerNorm(out_channel, data_format="channels_first")


class ConvNextBlock(nn.Module):
    def __init__(self, dim, kernel_size=7, layer_scale_init_value=1e-6):
        super().__init__()
        if kernel_size != 7:
            raise ValueError("cor_amd depthwise kernel is 7x7")
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=kernel_size, padding=kernel_size // 2, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim))


class GenerateMaskAdapterMap(nn.Module):
    def __init__(self, clip_in_channel=768, mask_downscaling_mid_channel=16, mid_channel=768, num_output_maps=16):
        super().__init__()
        self.clip_in_channel = clip_in_channel
        self.fuse = nn.Conv2d(clip_in_channel, mid_channel, 1)
        self.cnext1, self.cnext2, self.cnext3 = ConvNextBlock(mid_channel), ConvNextBlock(mid_channel), ConvNextBlock(mid_channel)
        self.norm = LayerNorm(mid_channel, data_format="channels_last")
        self.final = nn.Conv2d(mid_channel, num_output_maps, 1)
        m = mask_downscaling_mid_channel
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, m // 4, kernel_size=3, stride=2, padding=1), LayerNorm(m // 4, data_format="channels_first"), nn.GELU(),
            nn.Conv2d(m // 4, m, kernel_size=3, stride=2, padding=1), LayerNorm(m, data_format="channels_first"), nn.GELU(),
            nn.Conv2d(m, clip_in_channel, kernel_size=1))


class MaskAdapterPooling(nn.Module):
    def __init__(self, x_in_channel=1152, mask_adatpet_network_in_channel=256, mask_downscaling_mid_channel=16,
                 mask_adatpet_network_mid_channel=256, num_output_maps=16):
        super().__init__()
        self.channel_clip_to_maskadapter = ChannelReduction(x_in_channel, mask_adatpet_network_in_channel)
        self.get_mask_map = GenerateMaskAdapterMap(mask_adatpet_network_in_channel, mask_downscaling_mid_channel,
                                                   mask_adatpet_network_mid_channel, num_output_maps)
        self.num_output_maps = num_output_maps
