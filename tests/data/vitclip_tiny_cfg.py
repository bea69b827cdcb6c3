_base_ = ['./_base_/vitclip_tiny_model.py']
# per-dataset overrides, same shape as configs/recognition/vit/vitclip_base_k400.py:5-8
model = dict(
    backbone=dict(drop_path_rate=0.0, adapter_scale=0.5, num_frames=4),
    cls_head=dict(num_classes=7),
    test_cfg=dict(max_testing_views=4))
optimizer = dict(type='AdamW', lr=3e-4, betas=(0.9, 0.999), weight_decay=0.05,
                 paramwise_cfg=dict(custom_keys={'class_embedding': dict(decay_mult=0.),
                                                 'positional_embedding': dict(decay_mult=0.),
                                                 'ln_1': dict(decay_mult=0.),
                                                 'ln_2': dict(decay_mult=0.),
                                                 'ln_pre': dict(decay_mult=0.),
                                                 'ln_post': dict(decay_mult=0.)}))
optimizer_config = dict(type="DistOptimizerHook", update_interval=1, grad_clip=None, coalesce=True,
                        bucket_size_mb=-1, use_fp16=True)
