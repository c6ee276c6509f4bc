"""Hyper-parameters of the STYLER hot path (values of the reference's hparams.py; the kernels are
specialised for them: hidden 256, 4 heads x 64, FFN 1024 with k = (9, 1), predictors k = 3)."""
# Quantization for F0 and energy (hparams.py:21-25)
f0_min, f0_max = 71.0, 797.9
energy_min, energy_max = 0.1, 525.43

# Audio and mel (hparams.py:27-39)
sampling_rate = 22050
filter_length = 1024
hop_length = 256
win_length = 1024
n_bins = 256
max_wav_value = 32768.0
n_mel_channels = 80
mel_fmin, mel_fmax = 0.0, 8000.0

# STYLER (hparams.py:42-76)
n_src_vocab = 152                 # len(text.symbols) + 1 (transformer/Models.py:37)
encoder_layer, encoder_head, encoder_hidden = 2, 4, 256
decoder_layer, decoder_head, decoder_hidden = 4, 4, 256
fft_conv1d_filter_size = 1024
fft_conv1d_kernel_size = (9, 1)
encoder_dropout = decoder_dropout = 0.2
style_predictor_filter_size = 256
style_predictor_kernel_size = 3
style_predictor_dropout = 0.5
max_seq_len = 1000
dat_weight = 1
speaker_embed_dim = 512
va_neck_hidden_t, va_neck_hidden_r, va_neck_hidden_d, va_neck_hidden_p, va_neck_hidden_e = 4, 64, 80, 64, 64
va_enc_dim_r, va_enc_dim_d, va_enc_dim_p, va_enc_dim_e = 256, 256, 320, 320
va_dim_f0 = va_dim_energy = 257
va_chs_grp = 16

# Optimizer (hparams.py:93-101)
batch_size = 16
n_warm_up_step = 4000
grad_clip_thresh = 1.0
acc_steps = 1
betas = (0.9, 0.98)
eps = 1e-9
weight_decay = 0.0
log_offset = 1.0
