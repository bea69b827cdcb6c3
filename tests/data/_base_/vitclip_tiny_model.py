# model settings (tiny stand-in with the same keys as configs/_base_/models/vitclip_base.py)
model = dict(
    type='Recognizer3D',
    backbone=dict(
        type='ViT_CLIP',
        input_resolution=32,
        patch_size=16,
        num_frames=2,
        width=128,
        layers=2,
        heads=2,
        drop_path_rate=0.1),
    cls_head=dict(
        type='I3DHead',
        in_channels=128,
        num_classes=400,
        spatial_type='avg',
        dropout_ratio=0.5),
    test_cfg=dict(average_clips='prob'))
