"""Mesh visualisation glue -- mirror of ``simple_romp/vis_human/main.py`` for the Sim3DR renderer
(``setup_renderer`` :11-21, ``rendering_romp_bev_results`` :23-113, item 'mesh') and of
``vis_utils.mesh_color_left2right`` (:147-153).  The rasterization runs on the device (renderer.py here);
pyrender / open3d back-ends, bird / side views and the cv2 overlays are not part of the MI355X path."""
import numpy as np
import torch

from .renderer import Sim3DR

# vis_utils.py:128-141 -- the palette persons are coloured with, left to right in the image
color_table_default = np.array([
    [0.4, 0.6, 1], [0.8, 0.7, 1], [0.1, 0.9, 1], [0.8, 0.9, 1], [1, 0.6, 0.4], [1, 0.7, 0.8], [1, 0.9, 0.1],
    [1, 0.9, 0.8], [0.9, 1, 1], [0.9, 0.7, 0.4], [0.8, 0.7, 1], [0.8, 0.9, 1], [0.9, 0.3, 0.1], [0.7, 1, 0.6],
    [0.7, 0.4, 0.6], [0.3, 0.5, 1]])[:, ::-1]


def setup_renderer(name='sim3dr', **kwargs):
    if name != 'sim3dr':
        raise NotImplementedError("renderer '%s': only 'sim3dr' runs on the MI355X path" % name)
    return Sim3DR(**kwargs)


def mesh_color_left2right(trans, color_table=None):
    """Colour index = rank of the person's x translation (vis_utils.py:147-153)."""
    order = torch.sort(trans[:, 0].cpu()).indices.numpy()
    inds = np.arange(len(trans))
    inds[order] = np.arange(len(trans))
    table = color_table_default if color_table is None else color_table
    return np.array([table[i % len(table)] for i in inds])


def rendering_romp_bev_results(renderer, outputs, image, rendering_cfgs, alpha=1):
    """main.py:23-113 for renderer 'sim3dr', item 'mesh': persons painted far to near onto the frame;
    `outputs['rendered_image']` = [frame | rendering] side by side."""
    triangles = outputs['smpl_face'].cpu().numpy().astype(np.int32)
    cam_trans = outputs['cam_trans']
    if rendering_cfgs['mesh_color'] == 'identity':
        mesh_colors = mesh_color_left2right(cam_trans)
    elif rendering_cfgs['mesh_color'] == 'same':
        mesh_colors = np.array([[.9, .9, .8] for _ in range(len(cam_trans))])
    else:
        raise ValueError(rendering_cfgs['mesh_color'])
    unsupported = [it for it in rendering_cfgs['items'] if it != 'mesh']
    if unsupported:
        raise NotImplementedError('show_items %s need OpenCV drawing / extra views; only "mesh" is on the device path' % unsupported)
    result_image = [image]
    depth_order = torch.sort(cam_trans[:, 2].cpu(), descending=True).indices
    vertices = outputs['verts_camed_org'][depth_order.to(outputs['verts_camed_org'].device)].clone()
    vertices[:, :, 2] = vertices[:, :, 2] * -1
    rendered = renderer(vertices, triangles, np.ascontiguousarray(image), mesh_colors=mesh_colors[depth_order.numpy()])
    result_image.append(rendered)
    outputs['rendered_image'] = np.concatenate(result_image, 1)
    return outputs
